// Row maps of a ragged caption batch, built on the device in ONE launch (round 4).
//
// The text tower drops the padded token rows of a [B, L] caption batch (HF's BertModel computes them: huggingface_builder.py:16-17;
// nothing downstream reads them - clip.py:111-120 pools under the mask).  What it needs from the 0/1 attention mask:
//   idx [cap]       flat position b*L + l of every real token in raster order, then -1 up to the next multiple of the GEMM tile height
//   inv [B*L]       packed row of every position, -1 at padded positions
//   row_start [B+1] first packed row of every sequence (the attention kernels address the packed rows with it)
//   info [4]        {number of real tokens, 1 if some sequence has a real token behind a padded one, 1 if the real tokens did not fit
//                    `cap` rows or their number differs from `expect` (>= 0), rows written incl. the -1 padding}
// Until round 3 this was ~15 torch index kernels with two host reads in the middle (nonzero(), bool(holes)).  One block does it: a
// wave per sequence counts its real tokens, the block scans the B counts in LDS, a wave per sequence ranks its tokens with ballots.
// 512 x 77 captions: one launch of a few microseconds, nothing read by the host when the loader supplies the caption lengths.
#include "common.h"

#define RM_THREADS 1024
#define RM_MAX_B 8192

static __global__ __launch_bounds__(RM_THREADS) void ragged_maps_kernel(const int64_t* __restrict__ mask, int B, int L, int multiple, int cap,
                                                                        int expect, int* __restrict__ idx, int* __restrict__ inv,
                                                                        int* __restrict__ row_start, int* __restrict__ info) {
    __shared__ int cnt[RM_MAX_B + 1];
    __shared__ int wsum[RM_THREADS / WAVE];
    __shared__ int holes;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    constexpr int NW = RM_THREADS / WAVE;
    if (tid == 0) holes = 0;
    __syncthreads();
    // (1) real tokens per sequence; a mask is a prefix mask iff its last real position + 1 equals its count
    for (int b = wave; b < B; b += NW) {
        int c = 0, last = -1;
        for (int l = lane; l < L; l += WAVE) {
            const bool m = mask[(long)b * L + l] != 0;
            c += m ? 1 : 0;
            if (m) last = l;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            c += __shfl_xor(c, o);
            last = max(last, __shfl_xor(last, o));
        }
        if (lane == 0) {
            cnt[b] = c;
            if (last + 1 != c) atomicOr(&holes, 1);
        }
    }
    __syncthreads();
    // (2) exclusive scan of the B counts: each thread owns `per` consecutive sequences
    const int per = (B + RM_THREADS - 1) / RM_THREADS;
    const int b0 = min(tid * per, B), b1 = min(b0 + per, B);
    int mine = 0;
    for (int b = b0; b < b1; ++b) mine += cnt[b];
    int inc = mine;                                   // inclusive scan over the wave
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
        const int v = __shfl_up(inc, o);
        if (lane >= o) inc += v;
    }
    if (lane == WAVE - 1) wsum[wave] = inc;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    int run = base + inc - mine;
    int total = 0;
    for (int w = 0; w < NW; ++w) total += wsum[w];
    __syncthreads();                                  // every thread has read the counts it owns and the wave sums
    for (int b = b0; b < b1; ++b) {
        const int c = cnt[b];
        cnt[b] = run;
        run += c;
    }
    if (tid == 0) cnt[B] = total;
    __syncthreads();
    // (positions never reach past the caller's [cap, D] buffers: when the host-side token count the buffers were sized from is smaller than
    //  what the mask holds - reported through info[2] one step later - the surplus tokens are dropped here instead of indexed out of bounds)
    for (int b = tid; b <= B; b += RM_THREADS) row_start[b] = min(cnt[b], cap);
    // (3) rank of every real token inside its sequence -> idx / inv
    for (int b = wave; b < B; b += NW) {
        int at = cnt[b];
        for (int l0 = 0; l0 < L; l0 += WAVE) {
            const int l = l0 + lane;
            const bool m = l < L && mask[(long)b * L + l] != 0;
            const unsigned long long bal = __ballot(m);
            const int pos = at + __popcll(bal & ((1ull << lane) - 1ull));
            if (l < L) {
                inv[(long)b * L + l] = (m && pos < cap) ? pos : -1;
                if (m && pos < cap) idx[pos] = b * L + l;
            }
            at += __popcll(bal);
        }
    }
    // (4) -1 rows up to a multiple of the tile height (zero rows for the GEMMs; never put back), inside the caller's buffer
    const int want = min(cap, (total + multiple - 1) / multiple * multiple);
    for (int i = total + tid; i < want; i += RM_THREADS) idx[i] = -1;
    if (tid == 0) {
        info[0] = total;
        info[1] = holes;
        info[2] = (total > cap || (expect >= 0 && expect != total)) ? 1 : 0;
        info[3] = max(want, min(total, cap));
    }
}

extern "C" int simseg_ragged_maps(const int64_t* mask, int64_t B, int64_t L, int64_t multiple, int64_t cap, int64_t expect, int32_t* idx,
                                  int32_t* inv, int32_t* row_start, int32_t* info, void* stream) {
    SS_CHECK(mask && idx && inv && row_start && info, "ragged_maps: null pointer");
    SS_CHECK(B >= 1 && B <= RM_MAX_B, "ragged_maps: 1 <= B <= %d sequences per call (got %ld)", RM_MAX_B, (long)B);
    SS_CHECK(L >= 1 && B * L < (1ll << 31) && multiple >= 1 && cap >= 0, "ragged_maps: bad sizes");
    hipLaunchKernelGGL(ragged_maps_kernel, dim3(1), dim3(RM_THREADS), 0, (hipStream_t)stream, mask, (int)B, (int)L, (int)multiple, (int)cap,
                       (int)expect, idx, inv, row_start, info);
    SS_LAUNCH_CHECK("ragged_maps");
    return 0;
}
