// Fully connected CRF refinement of the zero-shot segmentation candidate maps on the device.
//
// Replaces `dense_crf` of the reference's tool (tools/seg_evaluation.py:31-54, called at :153 for every visited candidate class):
//     DenseCRF2D(W, H, 2); U = -log([1 - p, p] + 1e-8); addPairwiseGaussian(sxy=3, compat=3);
//     addPairwiseBilateral(sxy=40, srgb=13, rgbim, compat=10); inference(3); argmax
// i.e. Kraehenbuehl & Koltun's mean-field inference with Potts compatibilities and two Gaussian kernels, each applied through a
// permutohedral lattice (Adams, Baek & Davis 2010) with symmetric normalisation - restated for the CPU in oracle/crf_ref.py
// (pydensecrf itself is not available in this image; the restatement is pinned against an exact O(N^2) mean-field).
//
// HBM-bound byte / index work (no MFMA): per image the bilateral lattice is built once and shared by all candidate maps (channels); the
// spatial lattice depends on the image size only and is built once per call for the whole batch:
//   simplex   : one thread per pixel - elevate the feature vector, round to the remainder-0 point, rank, barycentric weights; the
//               d+1 simplex vertices are packed into one 64-bit key each and inserted into an open-addressing hash table (CAS)
//   assign    : every lattice point draws a dense id in the raster order of the pixels that created it; neighbours: 2 (d+1) hash look-ups per point
//   lists     : per lattice point the (pixel, vertex) entries that touch it (count, wave-scan segment allocation, fill), once
//   filter    : splat as a GATHER over those lists (plain loads, no float atomics), d+1 blur passes over the lattice points
//               (new = old + (n1 + n2) / 2, ping-pong buffers, row 0 = the zero "missing neighbour"), slice
//   mean field: Q1 = sigmoid(t1 - t0), t_l = -U_l + sum_k w_k n_k K_k(n_k Q_l); with two labels Q0 = 1 - Q1, so
//               K(n Q0) = K(n) - K(n Q1): ONE filter of the C class-1 channels per kernel and iteration, K(n) once per lattice.
// Scalar constants of the lattice filter (alpha, the un-normalised blur) cancel in n = 1 / sqrt(K 1): kept as in the library.
#include "common.h"

namespace {

constexpr unsigned long long CRF_EMPTY = ~0ull;
constexpr int CRF_MAXC = 8;

// 64-bit lattice keys: D coordinates of B bits each (offset binary) and, above them, the index of the image in the batch - the
// lattices of the images of a batch live side by side in ONE hash table / point list and never share a point.
template <int D> struct KeyBits { static constexpr int B = D <= 2 ? 16 : 10; static constexpr int IMG_BITS = 64 - D * B > 14 ? 14 : 64 - D * B; };

template <int D>
__device__ __forceinline__ unsigned long long crf_pack(const int (&k)[D], int img) {
    constexpr int B = KeyBits<D>::B;
    unsigned long long key = (unsigned long long)img << (D * B);
#pragma unroll
    for (int i = 0; i < D; ++i) key |= (unsigned long long)((unsigned)(k[i] + (1 << (B - 1))) & ((1u << B) - 1)) << (i * B);
    return key;
}
template <int D>
__device__ __forceinline__ int crf_unpack(unsigned long long key, int (&k)[D]) {
    constexpr int B = KeyBits<D>::B;
#pragma unroll
    for (int i = 0; i < D; ++i) k[i] = (int)((key >> (i * B)) & ((1u << B) - 1)) - (1 << (B - 1));
    return (int)(key >> (D * B));
}
__device__ __forceinline__ unsigned crf_hash(unsigned long long x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return (unsigned)x;
}
// Two-tier open-addressing table.  The big table must hold the worst case (every (pixel, vertex) pair a distinct point: 2 x entries
// slots, 512 MiB for a 16-image batch of 512^2), but a photograph's lattice has ~15 entries per point - and 25 M CAS operations
// scattered over 512 MiB are DRAM-latency bound.  So keys go to a SMALL front table first (entries / 4 slots: resident in the memory-side
// cache), probing at most CRF_PROBES slots there, and only spill into the big table when all of those are taken by other keys.  Nothing is
// ever removed, so the protocol is consistent: a key is in the front table iff a free slot lay within its probe window when it was
// inserted, and an EMPTY slot inside the window proves absence everywhere.  Slots are numbered front table first.
constexpr int CRF_PROBES = 8;
struct CrfTable { unsigned long long* keys; int* id; unsigned smask, bmask; };

// returns the slot of `key`, with bit 31 set when THIS call created the entry (exactly one caller per lattice point does)
__device__ __forceinline__ int crf_insert(const CrfTable& t, unsigned long long key) {
    const unsigned h = crf_hash(key);
    unsigned slot = h & t.smask;
#pragma unroll 1
    for (int p = 0; p < CRF_PROBES; ++p) {
        const unsigned long long prev = atomicCAS(t.keys + slot, CRF_EMPTY, key);
        if (prev == CRF_EMPTY) return (int)(slot | 0x80000000u);
        if (prev == key) return (int)slot;
        slot = (slot + 1) & t.smask;
    }
    const unsigned base = t.smask + 1;
    slot = (h >> 3) & t.bmask;
    for (;;) {
        const unsigned long long prev = atomicCAS(t.keys + base + slot, CRF_EMPTY, key);
        if (prev == CRF_EMPTY) return (int)((base + slot) | 0x80000000u);
        if (prev == key) return (int)(base + slot);
        slot = (slot + 1) & t.bmask;
    }
}
__device__ __forceinline__ int crf_find(const CrfTable& t, unsigned long long key) {
    const unsigned h = crf_hash(key);
    unsigned slot = h & t.smask;
#pragma unroll 1
    for (int p = 0; p < CRF_PROBES; ++p) {
        const unsigned long long cur = t.keys[slot];
        if (cur == key) return t.id[slot];
        if (cur == CRF_EMPTY) return -1;
        slot = (slot + 1) & t.smask;
    }
    const unsigned base = t.smask + 1;
    slot = (h >> 3) & t.bmask;
    for (;;) {
        const unsigned long long cur = t.keys[base + slot];
        if (cur == key) return t.id[base + slot];
        if (cur == CRF_EMPTY) return -1;
        slot = (slot + 1) & t.bmask;
    }
}

// Internal pixel order.  Every per-pixel array of the solver (lattice offsets, barycentric weights, Q, unaries, filter outputs, norms)
// is indexed by an INTERNAL pixel number; only the kernels that touch the caller's tensors (rgb, prob, mask, q_out) translate it.  With
// tw = W / 16 > 0 (H and W multiples of 16) the internal order is tile-major - 256 consecutive pixels are one 16 x 16 image tile - so a
// block's pixels are neighbours in BOTH image directions: the block-level de-duplication of the hash build and of the entry counts sees
// a fraction of the distinct lattice points that a 256 x 1 strip of a raster row touches (the bilateral kernel is 40 pixels wide: a
// strip crosses ~6 lattice cells, a tile stays inside one), the point ids drawn in creation order become 2-D local, and so do the
// gathers / slices / blur neighbours.  tw = 0: raster order (any image size).
__device__ __forceinline__ int crf_raster(int li, int W, int tw) {
    if (tw == 0) return li;
    const int t = li >> 8, w = li & 255;
    return ((t / tw) * 16 + (w >> 4)) * W + (t % tw) * 16 + (w & 15);
}

// Permutohedral::init, one thread per (internal) pixel.  D = 2: (x, y) / sxy;  D = 5: (x, y) / sxy, rgb / srgb.
template <int D>
__global__ __launch_bounds__(256) void crf_simplex_kernel(const unsigned char* __restrict__ rgb, int nimg, int H, int W, int tw, float inv_sxy,
                                                          float inv_srgb, CrfTable table, int* __restrict__ off, float* __restrict__ bary,
                                                          int* __restrict__ overflow) {
    // Block-level de-duplication in front of the global table: neighbouring pixels share most of their simplex vertices (a lattice point
    // has ~15 entries), and same-key CAS operations on one global address serialise.  The block's 256 x (D + 1) keys first meet in an LDS
    // table; only the thread that creates a key THERE goes to the global table and publishes the slot for the others.
    constexpr int LSLOTS = 2048;
    __shared__ unsigned long long lkeys[LSLOTS];
    __shared__ int lvals[LSLOTS];
    for (int t = threadIdx.x; t < LSLOTS; t += 256) lkeys[t] = CRF_EMPTY;
    __syncthreads();
    const long i0 = (long)blockIdx.x * 256 + threadIdx.x;         // pixel of the batch: image i / (H W)
    const bool valid = i0 < (long)nimg * H * W;
    const long i = valid ? i0 : (long)nimg * H * W - 1;
    const int img = (int)(i / (H * W)), li = crf_raster((int)(i % (H * W)), W, tw);
    float f[D];
    f[0] = (float)(li % W) * inv_sxy;
    f[1] = (float)(li / W) * inv_sxy;
    if constexpr (D == 5) {
#pragma unroll
        for (int c = 0; c < 3; ++c) f[2 + c] = (float)rgb[((long)img * H * W + li) * 3 + c] * inv_srgb;
    }
    const float inv_std_dev = sqrtf(2.0f / 3.0f) * (D + 1);
    float elevated[D + 1];
    float sm = 0.f;
#pragma unroll
    for (int j = D; j > 0; --j) {
        const float cf = f[j - 1] * (1.0f / sqrtf((float)(j * (j + 1))) * inv_std_dev);
        elevated[j] = sm - (float)j * cf;
        sm += cf;
    }
    elevated[0] = sm;
    const float down_factor = 1.0f / (D + 1), up_factor = (float)(D + 1);
    float rem0[D + 1];
    float fsum = 0.f;
#pragma unroll
    for (int k = 0; k <= D; ++k) {
        const float v = down_factor * elevated[k];
        const float up = ceilf(v) * up_factor, down = floorf(v) * up_factor;
        rem0[k] = (up - elevated[k] < elevated[k] - down) ? up : down;
        fsum += rem0[k] * down_factor;
    }
    const int isum = (int)rintf(fsum);
    int rank[D + 1];
#pragma unroll
    for (int k = 0; k <= D; ++k) rank[k] = 0;
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = a + 1; b <= D; ++b) {
            if (elevated[a] - rem0[a] < elevated[b] - rem0[b]) rank[a]++;
            else rank[b]++;
        }
#pragma unroll
    for (int k = 0; k <= D; ++k) {
        rank[k] += isum;
        if (rank[k] < 0) { rank[k] += D + 1; rem0[k] += up_factor; }
        else if (rank[k] > D) { rank[k] -= D + 1; rem0[k] -= up_factor; }
    }
    float bc[D + 2];
#pragma unroll
    for (int k = 0; k <= D + 1; ++k) bc[k] = 0.f;
#pragma unroll
    for (int k = 0; k <= D; ++k) {
        const float v = (elevated[k] - rem0[k]) * down_factor;
        // (rank is a small runtime value: select by comparison so the array stays in registers)
#pragma unroll
        for (int t = 0; t <= D + 1; ++t) {
            if (t == D - rank[k]) bc[t] += v;
            if (t == D + 1 - rank[k]) bc[t] -= v;
        }
    }
    bc[0] += 1.0f + bc[D + 1];
    bool bad = false;
    int ls[D + 1];                                                // LDS slot of each vertex key, bit 31: this thread created it there
#pragma unroll
    for (int r = 0; r <= D; ++r) {
        int key[D];
#pragma unroll
        for (int k = 0; k < D; ++k) {
            key[k] = (int)rintf(rem0[k]) + (rank[k] <= D - r ? r : r - (D + 1));        // canonical simplex, remainder r
            bad = bad || key[k] < -(1 << (KeyBits<D>::B - 1)) || key[k] >= (1 << (KeyBits<D>::B - 1));
        }
        const unsigned long long pk = crf_pack<D>(key, img);
        ls[r] = -1;
        if (valid) {
            unsigned slot = crf_hash(pk) & (LSLOTS - 1);
            for (;;) {
                const unsigned long long prev = atomicCAS(&lkeys[slot], CRF_EMPTY, pk);
                if (prev == CRF_EMPTY) { lvals[slot] = crf_insert(table, pk); ls[r] = (int)(slot | 0x80000000u); break; }   // global slot (+ creator bit)
                if (prev == pk) { ls[r] = (int)slot; break; }
                slot = (slot + 1) & (LSLOTS - 1);
            }
        }
    }
    __syncthreads();
    if (valid) {
#pragma unroll
        for (int r = 0; r <= D; ++r) {
            int g = lvals[ls[r] & 0x7fffffff];
            if (ls[r] >= 0) g &= 0x7fffffff;                      // the creator bit stays with the one entry that created the point
            off[(long)i * (D + 1) + r] = g;                       // hash slot (+ creator bit) for now; dense id after crf_assign
            bary[(long)i * (D + 1) + r] = bc[r];
        }
    }
    if (bad && valid) atomicExch(overflow, 1);
}

// Every lattice point draws a dense id, in the order of the (pixel, vertex) entries that CREATED the points (bit 31 of off[]): points
// are then numbered along the image raster, so the value rows a pixel's slice reads, the pixels a point's gather reads and most blur
// neighbours lie close together in memory (numbering in hash-slot order scattered all three over the whole array).  A thread looks at
// 16 consecutive entries.
// Round 5: the ids are EXACTLY in entry order - a count pass, an exclusive scan of the per-block totals and the assignment proper.  Rounds 2-4
// let every block draw its range from one atomic counter: blocks reach it in the order they happen to run, ~2000 of them (5 images' worth of
// pixels) in flight at any time, so ids 200 apart could lie images apart and a contiguous run of points touched pixels all over a
// 20 MB window - the gather's value reads missed L2 for 57 % of their requests (2.6 GB of fills per filter for 0.57 GB of operands,
// profiles/r5_crf_pmc.txt).  The numbering is now deterministic as well (the lattice no longer depends on scheduling).
__device__ __forceinline__ int crf_assign_local(const int* __restrict__ off, long nv, long e0, int (&slots)[16], int* wsum, int& n) {
    n = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        slots[j] = e0 + j < nv ? off[e0 + j] : 0;
        n += slots[j] < 0 ? 1 : 0;
    }
    int incl = n;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if ((int)(threadIdx.x & 63) >= o) incl += t;
    }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    return incl;
}
__global__ __launch_bounds__(256) void crf_assign_count_kernel(const int* __restrict__ off, long nv, int* __restrict__ btot) {
    __shared__ int wsum[4];
    int slots[16], n;
    crf_assign_local(off, nv, ((long)blockIdx.x * 256 + threadIdx.x) * 16, slots, wsum, n);
    if (threadIdx.x == 0) btot[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
// exclusive scan of the per-block totals in place (one block; a 32-image bilateral lattice has ~12 k blocks), total -> *M
__global__ __launch_bounds__(1024) void crf_assign_scan_kernel(int* __restrict__ btot, int nb, int* __restrict__ M) {
    __shared__ int part[1024];
    const int per = (nb + 1023) / 1024;
    const int b0 = threadIdx.x * per, b1 = min(nb, b0 + per);
    int sum = 0;
    for (int b = b0; b < b1; ++b) sum += btot[b];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {                           // Hillis-Steele over the 1024 partial sums
        const int t = (int)threadIdx.x >= o ? part[threadIdx.x - o] : 0;
        __syncthreads();
        part[threadIdx.x] += t;
        __syncthreads();
    }
    int run = part[threadIdx.x] - sum;
    for (int b = b0; b < b1; ++b) { const int c = btot[b]; btot[b] = run; run += c; }
    if (threadIdx.x == 1023) *M = part[1023];
}
__global__ __launch_bounds__(256) void crf_assign_kernel(const int* __restrict__ off, long nv, const unsigned long long* __restrict__ hkeys,
                                                         int* __restrict__ hid, unsigned long long* __restrict__ pkeys, const int* __restrict__ bbase) {
    __shared__ int wsum[4];
    int slots[16], n;
    const int incl = crf_assign_local(off, nv, ((long)blockIdx.x * 256 + threadIdx.x) * 16, slots, wsum, n);
    const int w = threadIdx.x >> 6;
    int id = bbase[blockIdx.x] + incl - n;
    for (int k = 0; k < w; ++k) id += wsum[k];
#pragma unroll
    for (int j = 0; j < 16; ++j)
        if (slots[j] < 0) {
            const int slot = slots[j] & 0x7fffffff;
            hid[slot] = id; pkeys[id] = hkeys[slot]; ++id;
        }
}

// ---- the splat as a gather: per lattice point the list of (pixel, vertex) entries that touch it, built once per lattice
// (count -> segment allocation -> fill), so that each of the ~10 filter applications of a CRF sums its values with plain loads
// instead of 6 (d = 5) float atomics per pixel and channel (the atomic splat was 1.8 ms per filter of a 16-image batch)
// rank[e] = position of entry e in its point's list, cnt[m] = list lengths.  A block's 1024 consecutive entries (~170 neighbouring
// pixels) mostly repeat a few hundred points: they are counted in an LDS table first and each distinct point costs the block ONE global
// atomicAdd (same-address global atomics serialise; this pass was 25 M of them).  The fill then needs no atomic at all.
// (hid: the entries still hold hash slots when this kernel starts - translated to dense point ids here, on the way)
__global__ __launch_bounds__(256) void crf_count_kernel(int* __restrict__ off, const int* __restrict__ hid, int* __restrict__ cnt, int* __restrict__ rank,
                                                        long nv) {
    // Round 5: the block's table is keyed by the HASH SLOT an entry still holds, and the slot -> dense id translation (a 4-byte read at a
    // random place of the hash-sized id array: a 128-byte L2 fill each) is made once per DISTINCT slot of the block instead of once per
    // entry - a block's 1024 entries hold a few hundred distinct points.
    constexpr int LSLOTS = 2048, EPT = 4;
    __shared__ int lid[LSLOTS], lcnt[LSLOTS], lbase[LSLOTS], lm[LSLOTS];
    for (int t = threadIdx.x; t < LSLOTS; t += 256) { lid[t] = -1; lcnt[t] = 0; }
    __syncthreads();
    const long e0 = (long)blockIdx.x * (256 * EPT) + threadIdx.x;
    int slot[EPT], lr[EPT];
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const long e = e0 + 256 * j;
        slot[j] = -1;
        if (e < nv) {
            const int hs = off[e] & 0x7fffffff;                   // global hash slot of the entry's point
            unsigned sl = ((unsigned)hs * 0x9E3779B1u) >> 21;      // 11 bits
            for (;;) {
                const int prev = atomicCAS(&lid[sl], -1, hs);
                if (prev == -1 || prev == hs) break;
                sl = (sl + 1) & (LSLOTS - 1);
            }
            slot[j] = (int)sl;
            lr[j] = atomicAdd(&lcnt[sl], 1);
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < LSLOTS; t += 256)
        if (lid[t] >= 0) {
            const int m = hid[lid[t]];
            lm[t] = m;
            lbase[t] = atomicAdd(cnt + m, lcnt[t]);
        }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < EPT; ++j)
        if (slot[j] >= 0) {
            off[e0 + 256 * j] = lm[slot[j]];                       // the entries hold dense point ids from here on
            rank[e0 + 256 * j] = lbase[slot[j]] + lr[j];
        }
}
__global__ __launch_bounds__(256) void crf_alloc_kernel(const int* __restrict__ cnt, int* __restrict__ start, const int* __restrict__ M,
                                                        int* __restrict__ cursor) {
    // Grid-stride over the M points that exist (the launch cannot know M: sized for the worst case it ran 6 threads per pixel, nearly all
    // idle), four points per thread, and ONE range per block from the single cursor word (27 k same-address atomics were the rest of this
    // kernel's 125 us per 16-image lattice).  Lists of consecutive points stay consecutive inside a block's range.
    __shared__ int wsum[4];
    __shared__ int bbase;
    const int Mv = *M;
    for (long b0 = (long)blockIdx.x * 1024; b0 < Mv; b0 += (long)gridDim.x * 1024) {
        const long m0 = b0 + threadIdx.x * 4;
        int c[4], n = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) { c[j] = m0 + j < Mv ? cnt[m0 + j] : 0; n += c[j]; }
        int incl = n;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o, 64);
            if ((int)(threadIdx.x & 63) >= o) incl += t;
        }
        const int w = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 63) wsum[w] = incl;
        __syncthreads();
        if (threadIdx.x == 0) {
            const int tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
            bbase = tot ? atomicAdd(cursor, tot) : 0;
        }
        __syncthreads();
        int s = bbase + incl - n;
        for (int k = 0; k < w; ++k) s += wsum[k];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (m0 + j < Mv) { start[m0 + j] = s; s += c[j]; }
        __syncthreads();                                          // wsum / bbase are rewritten by the next trip
    }
}
__global__ __launch_bounds__(256) void crf_fill_kernel(const int* __restrict__ off, const float* __restrict__ bary, const int* __restrict__ start,
                                                       const int* __restrict__ rank, int2* __restrict__ rec, long nv, int d1) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= nv) return;
    const int m = off[e];
    const int slot = start[m] + rank[e];
    rec[slot] = make_int2((int)(e / d1), __float_as_int(bary[e]));     // {pixel, barycentric weight}: ONE 8-byte record per entry
}
// val[(m + 1) * CS + c] = sum over the entries e of point m of w_e * in[pixel_e * CS + c]      (w_e already times scale[pixel] in `recs`)
// G lanes per lattice point walk its list together (a point of the bilateral lattice has ~10-25 entries, of the spatial one more): the
// list reads are contiguous per group instead of one cache line per lane, and the partial sums meet in a shuffle tree.
// CS = channel stride of every per-pixel / per-point array of the solver (4 for up to 4 channels, else 8): an entry costs one 8-byte
// record load and one (two) 16-byte value load(s) instead of 2 + C scalar loads - the scattered reads are sector-sized either way.
template <int D, int CS>
__global__ __launch_bounds__(256) void crf_gather_kernel(const float* __restrict__ in, const int* __restrict__ start, const int* __restrict__ cnt,
                                                         const int2* __restrict__ rec, float* __restrict__ val, const int* __restrict__ M, int C,
                                                         long in_stride, long val_stride, int extra) {
    // extra = 1: channel C (behind the C channels of `in`) has the constant input 1 (times the entry weight): the filter of the lattice's
    // norm rides along with the first mean-field filter instead of being a filter application of its own
    constexpr int G = D == 2 ? 16 : 8;
    const int sub = threadIdx.x & (G - 1);
    if (in) in += (long)blockIdx.y * in_stride;                   // blockIdx.y: images that SHARE this lattice (the spatial one), see crf_filter
    val += (long)blockIdx.y * val_stride;
    const long Mr = ((long)*M + 256 / G - 1) / (256 / G) * (256 / G);            // whole groups stay together through the shuffles
    // Block -> point map (round 5): every block owns ONE contiguous run of points, and the blocks of an XCD (blockIdx % 8, xcd_remap) own
    // neighbouring runs.  Point ids follow the tile-major pixel raster, and a pixel's value row is read through up to d + 1 points with
    // nearby ids: with a grid-stride map those reads were spread over all eight XCDs' L2s (each a miss of its own: 128-byte fills for 16
    // useful bytes); now they meet in one L2.
    constexpr int PPB = 256 / G;                                  // points per block and trip
    const long trips = (Mr / PPB + gridDim.x - 1) / gridDim.x;    // trips of every block over its run
    const long run0 = (long)xcd_remap(blockIdx.x, gridDim.x) * trips * PPB;
    for (long t = 0, m = run0 + threadIdx.x / G; t < trips && m < Mr; ++t, m += PPB) {
        float acc[CS];
#pragma unroll
        for (int c = 0; c < CS; ++c) acc[c] = 0.f;
        const bool live = m < *M;
        const int s0 = live ? start[m] : 0, n = live ? cnt[m] : 0;
        // U entries per lane and trip, every record requested before the first value, every value before the first use: a list of up to
        // U * G entries (the typical bilateral point has 10-25) costs two memory round trips instead of two per entry of a lane (round 5)
        constexpr int U = CS == 4 ? 4 : 2;
        for (int j0 = sub; j0 < n; j0 += U * G) {
            int2 r[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = j0 + u * G;
                r[u] = j < n ? rec[s0 + j] : make_int2(0, 0);            // weight 0: the entry contributes nothing (pixel 0 is a valid address)
            }
            float v[U][CS];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (in) {
#pragma unroll
                    for (int q = 0; q < CS / 4; ++q) {
                        const float4 t = *reinterpret_cast<const float4*>(in + (long)r[u].x * CS + 4 * q);
                        v[u][4 * q] = t.x; v[u][4 * q + 1] = t.y; v[u][4 * q + 2] = t.z; v[u][4 * q + 3] = t.w;
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < CS; ++c) v[u][c] = 1.0f;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float w = __int_as_float(r[u].y);
#pragma unroll
                for (int c = 0; c < CS; ++c) acc[c] += w * ((extra && c == C) ? 1.0f : v[u][c]);
            }
        }
        // the group's partial sums meet through the DPP network (round 5; was a ds_bpermute tree: 3-4 LDS-crossbar round trips per channel):
        // quad butterflies, then the half-row mirror (8 lanes) and, for 16-lane groups, the row mirror - every lane of the group ends with the sum
#pragma unroll
        for (int c = 0; c < CS; ++c) {
#define CRF_DPP_ADD(ctrl) acc[c] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc[c]), ctrl, 0xf, 0xf, true))
            CRF_DPP_ADD(0xB1);                        // quad_perm [1,0,3,2]
            CRF_DPP_ADD(0x4E);                        // quad_perm [2,3,0,1]
            CRF_DPP_ADD(0x141);                       // row_half_mirror: lanes i <-> 7 - i of every 8
            if (G == 16) CRF_DPP_ADD(0x140);          // row_mirror: lanes i <-> 15 - i of every 16
#undef CRF_DPP_ADD
        }
        if (live && sub == 0) {
#pragma unroll
            for (int q = 0; q < CS / 4; ++q)
                *reinterpret_cast<float4*>(val + (m + 1) * CS + 4 * q) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
        }
    }
}

// recs = {pixel, weight * norm[pixel]}: every filter of the mean field scales its input by the lattice's norm - folded into the list weights
// once, so the gather makes one scattered read per entry (the value) instead of two
__global__ __launch_bounds__(256) void crf_prescale_kernel(const int2* __restrict__ rec, const float* __restrict__ norm, int2* __restrict__ recs, long nv) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e < nv) {
        const int2 r = rec[e];
        recs[e] = make_int2(r.x, __float_as_int(__int_as_float(r.y) * norm[r.x]));
    }
}

template <int D>
__global__ __launch_bounds__(256) void crf_neighbors_kernel(const unsigned long long* __restrict__ pkeys, const int* __restrict__ M,
                                                            CrfTable table, int* __restrict__ nb, long mmax) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < *M; i += (long)gridDim.x * 256) {
        int key[D];
        const int img = crf_unpack<D>(pkeys[i], key);
#pragma unroll
        for (int j = 0; j <= D; ++j) {
            int k1[D], k2[D];
#pragma unroll
            for (int k = 0; k < D; ++k) { k1[k] = key[k] - 1; k2[k] = key[k] + 1; }
            if (j < D) { k1[j] = key[j] + D; k2[j] = key[j] - D; }
            // The two neighbours along an axis are each other's inverse (k2 of k1(i) is i): ONE hash search per point and axis, the found
            // point gets this one as its "minus" neighbour by a plain store (round 5; was two searches: 12 instead of 6 scattered probe
            // chains per point of the bilateral lattice, 68 M L2 misses per build).  crf_neighbors_init_kernel has set every "minus"
            // entry to -1 = "missing" beforehand.
            (void)k2;
            const int pn = crf_find(table, crf_pack<D>(k1, img));
            nb[((long)j * mmax + i) * 2 + 0] = pn;
            if (pn >= 0) nb[((long)j * mmax + pn) * 2 + 1] = (int)i;
        }
    }
}
__global__ __launch_bounds__(256) void crf_neighbors_init_kernel(const int* __restrict__ M, int* __restrict__ nb, long mmax, int d1) {
    const long total = (long)*M * d1;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const long j = t / *M, i = t - j * *M;
        nb[(j * mmax + i) * 2 + 1] = -1;
    }
}

template <int CS>
__global__ __launch_bounds__(256) void crf_blur_kernel(const float* __restrict__ src, float* __restrict__ dst, const int* __restrict__ nbj,
                                                       const int* __restrict__ M, long val_stride) {
    src += (long)blockIdx.y * val_stride;
    dst += (long)blockIdx.y * val_stride;
    const long total = (long)*M * (CS / 4);                       // one thread per point and 4 channels (the launch is sized for a typical lattice)
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const long i = t / (CS / 4);
        const int q = (int)(t % (CS / 4)) * 4;
        const long n1 = nbj[i * 2] + 1, n2 = nbj[i * 2 + 1] + 1;              // -1 -> row 0 (zeros)
        const float4 a = *reinterpret_cast<const float4*>(src + (i + 1) * CS + q);
        const float4 b = *reinterpret_cast<const float4*>(src + n1 * CS + q);
        const float4 c = *reinterpret_cast<const float4*>(src + n2 * CS + q);
        *reinterpret_cast<float4*>(dst + (i + 1) * CS + q) =
            make_float4(a.x + 0.5f * (b.x + c.x), a.y + 0.5f * (b.y + c.y), a.z + 0.5f * (b.z + c.z), a.w + 0.5f * (b.w + c.w));
    }
}

// slice: out[i * CS + c] = scale_i * alpha * sum_r bary * val[(off + 1) * CS + c]
template <int D, int CS>
__global__ __launch_bounds__(256) void crf_slice_kernel(const float* __restrict__ val, const float* __restrict__ scale, const int* __restrict__ off,
                                                        const float* __restrict__ bary, float* __restrict__ out, long N, int C, int sqrt_norm,
                                                        long val_stride, long out_stride, float* __restrict__ out2) {
    // out2 != nullptr: channel C of the value rows is the filtered norm (see crf_gather_kernel) and goes to out2[i].  out == the [pixel][CS]
    // array of the filter's result, or - when C == 1 and sqrt_norm / a [pixel] output is wanted (the lattice norms) - a plain [pixel] array
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    val += (long)blockIdx.y * val_stride;
    out += (long)blockIdx.y * out_stride;
    const float alpha = 1.0f / (1.0f + exp2f(-(float)D));
    float acc[CS];
#pragma unroll
    for (int c = 0; c < CS; ++c) acc[c] = 0.f;
#pragma unroll
    for (int r = 0; r <= D; ++r) {
        const long o = (long)(off[(long)i * (D + 1) + r] + 1) * CS;
        const float w = bary[(long)i * (D + 1) + r] * alpha;
#pragma unroll
        for (int q = 0; q < CS / 4; ++q) {
            const float4 t = *reinterpret_cast<const float4*>(val + o + 4 * q);
            acc[4 * q] += w * t.x; acc[4 * q + 1] += w * t.y; acc[4 * q + 2] += w * t.z; acc[4 * q + 3] += w * t.w;
        }
    }
    const float sc = scale ? scale[i] : 1.0f;
    if (out_stride == 0) {                                        // a per-pixel scalar output (the lattice norm / kn): channel 0
        float v = acc[0] * sc;
        if (sqrt_norm) v = 1.0f / sqrtf(v + 1e-20f);              // norm = 1 / sqrt(K 1 + 1e-20)
        out[i] = v;
        return;
    }
#pragma unroll
    for (int q = 0; q < CS / 4; ++q)
        *reinterpret_cast<float4*>(out + i * CS + 4 * q) = make_float4(acc[4 * q] * sc, acc[4 * q + 1] * sc, acc[4 * q + 2] * sc, acc[4 * q + 3] * sc);
    if (out2 && blockIdx.y == 0) out2[i] = acc[C] * sc;           // (images that share the lattice produce the same value)
}

// U = -log([1 - p, p] + 1e-8);  Q = softmax(-U)   (prob [C][N] -> q1 [N][C], u [N][C][2])
__global__ __launch_bounds__(256) void crf_init_kernel(const float* __restrict__ prob, float* __restrict__ q1, float* __restrict__ u, long NT, int N,
                                                       int C, int CS, int W, int tw) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;          // NT = images x N pixels of the batch; prob is [image][C][N]
    if (t >= NT * C) return;
    const long i = t / C;
    const int c = (int)(t % C);
    const long e = i * CS + c;                                    // q1 [pixel][CS], u [pixel][CS][2]
    const float p = prob[((i / N) * C + c) * N + crf_raster((int)(i % N), W, tw)];
    const float u0 = -logf((1.0f - p) + 1e-8f), u1 = -logf(p + 1e-8f);
    u[e * 2] = u0; u[e * 2 + 1] = u1;
    const float t0 = -u0, t1 = -u1, mx = fmaxf(t0, t1);
    const float e0 = expf(t0 - mx), e1 = expf(t1 - mx);
    q1[e] = e1 / (e0 + e1);
}

// t_l = -U_l + wg * fg_l + wb * fb_l with f_1 = filtered Q1, f_0 = K(n) n - f_1;  Q1 = softmax;  last iteration: mask = 255 * (t1 > t0)
__global__ __launch_bounds__(256) void crf_update_kernel(const float* __restrict__ u, const float* __restrict__ fg, const float* __restrict__ fb,
                                                         const float* __restrict__ kng, const float* __restrict__ knb, float wg, float wb,
                                                         float* __restrict__ q1, unsigned char* __restrict__ mask, float* __restrict__ q_out,
                                                         long NT, int N, int C, int CS, int W, int tw) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= NT * C) return;
    const long i = t / C;
    const int c = (int)(t % C);
    const long e = i * CS + c;
    const float g1 = fg[e], b1 = fb[e];
    const float t0 = -u[e * 2] + wg * (kng[i % N] - g1) + wb * (knb[i] - b1);     // (the spatial lattice is one image's, shared)
    const float t1 = -u[e * 2 + 1] + wg * g1 + wb * b1;
    const float mx = fmaxf(t0, t1);
    const float e0 = expf(t0 - mx), e1 = expf(t1 - mx);
    const float q = e1 / (e0 + e1);
    q1[e] = q;
    const long o = ((i / N) * C + c) * N + crf_raster((int)(i % N), W, tw);
    if (mask) mask[o] = (e1 / (e0 + e1) > e0 / (e0 + e1)) ? 255 : 0;                          // argmax over [Q0, Q1], ties -> label 0
    if (q_out) q_out[o] = q;
}

struct CrfLattice {
    int* off; float* bary; unsigned long long* hkeys; int* hid; unsigned long long* pkeys; int* nb; float* norm; float* kn; int* M;
    int* cnt; int* start; int2* rec; int2* recs; int* cursor;   // per-point entry lists: {pixel, weight} (and {pixel, weight x norm[pixel]}) of every (pixel, vertex) pair
    long cap, scap, mmax;             // big / front table slots (hkeys, hid hold scap + cap entries)
};

struct CrfLayout {
    size_t total = 0;
    template <typename T> T* carve(char* base, size_t n) {
        total = (total + 255) & ~size_t(255);
        T* p = base ? reinterpret_cast<T*>(base + total) : nullptr;
        total += n * sizeof(T);
        return p;
    }
};

long crf_cap(long n) {
    long c = 1024;
    while (c < 2 * n) c <<= 1;
    return c;
}

// the same carving serves the size query (base == nullptr) and the launch
void crf_carve(CrfLayout& L, char* base, long N, long N1, int C, CrfLattice (&lat)[2], float*& val0, float*& val1, float*& q1, float*& u, float*& fg,
               float*& fb, int*& flags) {
    const int dims[2] = {2, 5};
    const long npx[2] = {N1, N};              // the spatial lattice is one image's (N1 pixels), the bilateral one spans the batch (N)
    for (int k = 0; k < 2; ++k) {
        const long nv = npx[k] * (dims[k] + 1);
        lat[k].cap = crf_cap(nv);
        lat[k].scap = crf_cap(nv / 8);        // front table: entries / 4 slots
        lat[k].mmax = nv;
        lat[k].off = L.carve<int>(base, nv);
        lat[k].bary = L.carve<float>(base, nv);
        lat[k].hkeys = L.carve<unsigned long long>(base, lat[k].scap + lat[k].cap);
        lat[k].hid = L.carve<int>(base, lat[k].scap + lat[k].cap);
        lat[k].pkeys = L.carve<unsigned long long>(base, nv);
        lat[k].nb = L.carve<int>(base, nv * (dims[k] + 1) * 2);
        lat[k].norm = L.carve<float>(base, npx[k]);
        lat[k].kn = L.carve<float>(base, npx[k]);
        lat[k].cnt = L.carve<int>(base, 2 * nv);              // counts per point, then the rank of every entry in its point's list
        lat[k].start = L.carve<int>(base, nv);
        lat[k].rec = L.carve<int2>(base, nv);
        lat[k].recs = L.carve<int2>(base, nv);
        // the lattice's own point count and list cursor (64 ints: with everything above, the spatial lattice's region depends on the image
        // size only - a later call on the same workspace can reuse it, see reuse_spatial)
        lat[k].M = L.carve<int>(base, 64);
        lat[k].cursor = lat[k].M ? lat[k].M + 1 : nullptr;
    }
    const long CS = C + 1 <= 4 ? 4 : 8;                          // channel stride (the C maps + the norm channel of the first filter, padded)
    const long vmax = (N * 6 + 1) * CS;
    val0 = L.carve<float>(base, vmax);
    val1 = L.carve<float>(base, vmax);
    q1 = L.carve<float>(base, N * CS);
    u = L.carve<float>(base, N * CS * 2);
    fg = L.carve<float>(base, N * CS);
    fb = L.carve<float>(base, N * CS);
    flags = L.carve<int>(base, 8);            // [2] = key overflow flag of the hash builds
}

unsigned crf_blocks(long work) {
    long b = (work + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

// One filter application K(scale_in * in) * scale_out over `nimg` images that share the lattice `lt` (N pixels each; in / out are
// [image][pixel][C], scale arrays [pixel]): the bilateral lattice covers the whole batch (nimg = 1, N = all pixels), the spatial one is
// one image's and serves every image of the batch through blockIdx.y.
template <int D, int CS>
void crf_filter_cs(const CrfLattice& lt, const float* in, bool scaled_in, const float* scale_out, float* out, bool scalar_out, float* val0, float* val1,
                   long N, int C, int sqrt_norm, int nimg, hipStream_t s, float* out2) {
    const long vstride = (N * (D + 1) + 1) * (long)CS;
    (void)hipMemset2DAsync(val0, vstride * sizeof(float), 0, (size_t)CS * sizeof(float), nimg, s);   // row 0 = the zero "missing neighbour"
    (void)hipMemset2DAsync(val1, vstride * sizeof(float), 0, (size_t)CS * sizeof(float), nimg, s);
    hipLaunchKernelGGL((crf_gather_kernel<D, CS>), dim3(crf_blocks(N / 4), nimg), dim3(256), 0, s, in, lt.start, lt.cnt, scaled_in ? lt.recs : lt.rec, val0,
                       lt.M, C, N * CS, vstride, out2 ? 1 : 0);
    float* a = val0;
    float* b = val1;
    for (int j = 0; j <= D; ++j) {
        // lattice points are a fraction of the worst case N (D + 1): a grid-stride launch sized for N / 4 points
        hipLaunchKernelGGL(crf_blur_kernel<CS>, dim3(crf_blocks(N / 4 * (CS / 4)), nimg), dim3(256), 0, s, a, b, lt.nb + (long)j * lt.mmax * 2, lt.M, vstride);
        float* t = a; a = b; b = t;
    }
    hipLaunchKernelGGL((crf_slice_kernel<D, CS>), dim3((unsigned)((N + 255) / 256), nimg), dim3(256), 0, s, a, scale_out, lt.off, lt.bary, out, N, C, sqrt_norm,
                       vstride, scalar_out ? 0 : N * CS, out2);
}

// One filter application K(scale_in * in) * scale_out over `nimg` images that share the lattice `lt` (N pixels each; in / out are
// [image][pixel][CS] with CS = 4 or 8 the padded channel count, or - scalar_out - a plain [pixel] array from channel 0; scale arrays
// [pixel]): the bilateral lattice covers the whole batch (nimg = 1, N = all pixels), the spatial one is one image's and serves every image of
// the batch through blockIdx.y.  in == nullptr: the constant 1.  out2: one more channel, the filter of the constant 1 (-> the lattice's kn).
template <int D>
void crf_filter(const CrfLattice& lt, const float* in, bool scaled_in, const float* scale_out, float* out, bool scalar_out, float* val0, float* val1, long N,
                int C, int CS, int sqrt_norm, int nimg, hipStream_t s, float* out2 = nullptr) {
    if (CS == 4) crf_filter_cs<D, 4>(lt, in, scaled_in, scale_out, out, scalar_out, val0, val1, N, C, sqrt_norm, nimg, s, out2);
    else crf_filter_cs<D, 8>(lt, in, scaled_in, scale_out, out, scalar_out, val0, val1, N, C, sqrt_norm, nimg, s, out2);
}

template <int D>
void crf_build(const CrfLattice& lt, const unsigned char* rgb, int nimg, int H, int W, int tw, float sxy, float srgb, int* overflow, float* val0,
               float* val1, hipStream_t s, bool with_kn) {
    const long N = (long)nimg * H * W;
    const long nv = N * (D + 1);
    (void)hipMemsetAsync(lt.M, 0, 2 * sizeof(int), s);          // point count + list cursor
    (void)hipMemsetAsync(lt.hkeys, 0xFF, (size_t)(lt.scap + lt.cap) * sizeof(unsigned long long), s);
    const CrfTable table{lt.hkeys, lt.hid, (unsigned)(lt.scap - 1), (unsigned)(lt.cap - 1)};
    hipLaunchKernelGGL(crf_simplex_kernel<D>, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, rgb, nimg, H, W, tw, 1.0f / sxy, 1.0f / srgb, table,
                       lt.off, lt.bary, overflow);
    {   // dense point ids in entry order: count, scan, assign (the per-block totals borrow the head of lt.cnt, which is zeroed afterwards)
        const unsigned nb = (unsigned)((nv + 4095) / 4096);
        hipLaunchKernelGGL(crf_assign_count_kernel, dim3(nb), dim3(256), 0, s, lt.off, nv, lt.cnt);
        hipLaunchKernelGGL(crf_assign_scan_kernel, dim3(1), dim3(1024), 0, s, lt.cnt, (int)nb, lt.M);
        hipLaunchKernelGGL(crf_assign_kernel, dim3(nb), dim3(256), 0, s, lt.off, nv, lt.hkeys, lt.hid, lt.pkeys, lt.cnt);
    }
    hipLaunchKernelGGL(crf_neighbors_init_kernel, dim3(crf_blocks(N / 4 * (D + 1))), dim3(256), 0, s, lt.M, lt.nb, lt.mmax, D + 1);
    hipLaunchKernelGGL(crf_neighbors_kernel<D>, dim3(crf_blocks(N / 4)), dim3(256), 0, s, lt.pkeys, lt.M, table, lt.nb, lt.mmax);
    (void)hipMemsetAsync(lt.cnt, 0, (size_t)nv * sizeof(int), s);
    hipLaunchKernelGGL(crf_count_kernel, dim3((unsigned)((nv + 1023) / 1024)), dim3(256), 0, s, lt.off, lt.hid, lt.cnt, lt.cnt + nv, nv);
    hipLaunchKernelGGL(crf_alloc_kernel, dim3(crf_blocks(N / 4)), dim3(256), 0, s, lt.cnt, lt.start, lt.M, lt.cursor);       // (1024 points per block and trip)
    hipLaunchKernelGGL(crf_fill_kernel, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, s, lt.off, lt.bary, lt.start, lt.cnt + nv, lt.rec, nv, D + 1);
    // norm = 1 / sqrt(K 1 + 1e-20);  kn = norm * K(norm)   (the filtered constant-one channel of the symmetric normalisation)
    crf_filter<D>(lt, nullptr, false, nullptr, lt.norm, true, val0, val1, N, 1, 4, 1, 1, s);
    hipLaunchKernelGGL(crf_prescale_kernel, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, s, lt.rec, lt.norm, lt.recs, nv);
    // (with_kn = false: the caller's first mean-field filter carries the constant-one channel along and writes kn itself)
    if (with_kn) crf_filter<D>(lt, nullptr, true, lt.norm, lt.kn, true, val0, val1, N, 1, 4, 0, 1, s);
}

// largest |lattice coordinate| the features can produce.  Permutohedral::init: cf_i = f_i * scale_i and
//   elevated_j = sum_{i >= j} cf_i - j cf_{j-1}      (elevated_0 = sum_i cf_i, elevated_D = -D cf_{D-1}).
// The features here are pixel coordinates and colours, i.e. 0 <= f_i <= fmax_i, so the two parts never add up in magnitude:
//   |elevated_j| <= max(sum_{i >= j} fmax_i scale_i,  j fmax_{j-1} scale_{j-1})
// (round 4 bounded it by the SUM of the two, which refused 1024-pixel-wide images - 662 against the 512 the 10-bit keys of the bilateral
// lattice hold - although their coordinates stay below 220; the build kernels check every key they pack as well: flags[2]).
// + 2 (D + 1) for the rounding to the nearest remainder-0 point and the rank adjustments.
float crf_key_bound(int D, const float* fmax) {
    const float inv_std_dev = sqrtf(2.0f / 3.0f) * (D + 1);
    float cf[8], tail = 0.f, mx = 0.f;
    for (int i = 0; i < D; ++i) cf[i] = fmax[i] * inv_std_dev / sqrtf((float)((i + 1) * (i + 2)));
    for (int j = D; j >= 0; --j) {
        if (j < D) tail += cf[j];
        const float neg = j > 0 ? j * cf[j - 1] : 0.f;
        const float b = tail > neg ? tail : neg;
        mx = b > mx ? b : mx;
    }
    return mx + 2 * (D + 1);
}

}  // namespace

extern "C" int64_t simseg_dense_crf_workspace_bytes(int64_t B, int64_t H, int64_t W, int64_t C) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || C > CRF_MAXC) return -1;
    CrfLayout L;
    CrfLattice lat[2];
    float *v0, *v1, *q1, *u, *fg, *fb;
    int* flags;
    crf_carve(L, nullptr, B * H * W, H * W, (int)C, lat, v0, v1, q1, u, fg, fb, flags);
    return (int64_t)L.total + 256;
}

// rgb [B,H,W,3] uint8 (the de-normalised network inputs, RGB), prob [B,C,H,W] fp32 in [0,1] (min-max normalised candidate maps; the C
// maps of an image share its lattices) -> mask [B,C,H,W] uint8 (255 where the CRF labels the pixel as the class, else 0); q_out
// (optional) [B,C,H,W] = Q(label 1).  The images of a batch are independent problems solved side by side in the same launches.
extern "C" int simseg_dense_crf(const uint8_t* rgb, const float* prob, uint8_t* mask, float* q_out, int64_t B, int64_t C, int64_t H, int64_t W,
                                float sxy_g, float compat_g, float sxy_b, float srgb, float compat_b, int iters, void* workspace,
                                int64_t workspace_bytes, int reuse_spatial, void* stream) {
    SS_CHECK(rgb && prob && mask && workspace, "dense_crf: null pointer");
    SS_CHECK(B >= 1 && B < (1 << 14), "dense_crf: 1..16383 images per call (got %lld)", (long long)B);
    SS_CHECK(H > 0 && W > 0 && B * H * W < (1ll << 26), "dense_crf: batch of %lld images of %lld x %lld pixels is out of range", (long long)B,
             (long long)H, (long long)W);
    SS_CHECK(C >= 1 && C <= CRF_MAXC, "dense_crf: 1..%d candidate maps per image (got %lld)", CRF_MAXC, (long long)C);
    SS_CHECK(iters >= 1 && sxy_g > 0.f && sxy_b > 0.f && srgb > 0.f, "dense_crf: bad parameters");
    SS_CHECK(((uintptr_t)workspace % 256) == 0, "dense_crf: workspace must be 256-byte aligned");
    SS_CHECK(workspace_bytes >= simseg_dense_crf_workspace_bytes(B, H, W, C), "dense_crf: workspace too small (%lld < %lld bytes)",
             (long long)workspace_bytes, (long long)simseg_dense_crf_workspace_bytes(B, H, W, C));
    {   // lattice coordinates must fit the packed keys
        const float ext = (float)(H > W ? H : W);
        const float f2[2] = {ext / sxy_g, ext / sxy_g}, f5[5] = {ext / sxy_b, ext / sxy_b, 255.f / srgb, 255.f / srgb, 255.f / srgb};
        SS_CHECK(crf_key_bound(2, f2) < (float)(1 << 15) && crf_key_bound(5, f5) < (float)(1 << 9),
                 "dense_crf: feature range too large for the packed lattice keys (%g, %g)", crf_key_bound(2, f2), crf_key_bound(5, f5));
    }
    hipStream_t s = (hipStream_t)stream;
    const long NT = B * H * W;
    const int N = (int)(H * W), Ci = (int)C;
    CrfLayout L;
    CrfLattice lat[2];
    float *val0, *val1, *q1, *u, *fg, *fb;
    int* flags;
    crf_carve(L, static_cast<char*>(workspace), NT, N, Ci, lat, val0, val1, q1, u, fg, fb, flags);
    (void)hipMemsetAsync(flags, 0, 8 * sizeof(int), s);
    // the spatial lattice depends on (H, W, sxy) only: built for ONE image, shared by the batch
    const int tw = (H % 16 == 0 && W % 16 == 0) ? (int)(W / 16) : 0;       // tile-major internal pixel order where the image tiles evenly
    // K(n) of each lattice - the filter of its norm, needed by the first update - is one more channel of the first iteration's filter
    // (4 filter applications per lattice and call instead of 5) whenever the C candidate channels leave room for it
    // reuse_spatial: the caller vouches that the LAST call on this workspace had the same H, W and sxy_g and that nothing has written to the
    // workspace since - the spatial lattice (hash tables, neighbour lists, entry lists, norm and K(norm)) is still there and is not rebuilt
    const bool ride = Ci + 1 <= CRF_MAXC;
    if (!reuse_spatial) crf_build<2>(lat[0], rgb, 1, (int)H, (int)W, tw, sxy_g, 1.0f, flags + 2, val0, val1, s, !ride);
    crf_build<5>(lat[1], rgb, (int)B, (int)H, (int)W, tw, sxy_b, srgb, flags + 2, val0, val1, s, !ride);
    const long nc = NT * Ci;
    const int CS = Ci + 1 <= 4 ? 4 : 8;                          // as in crf_carve
    (void)hipMemsetAsync(q1, 0, (size_t)NT * CS * sizeof(float), s);          // (the padding channels are read - and ignored - by the vector loads)
    hipLaunchKernelGGL(crf_init_kernel, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, s, prob, q1, u, NT, N, Ci, CS, (int)W, tw);
    for (int it = 0; it < iters; ++it) {
        const bool first = ride && it == 0;
        crf_filter<2>(lat[0], q1, true, lat[0].norm, fg, false, val0, val1, N, Ci, CS, 0, (int)B, s, (first && !reuse_spatial) ? lat[0].kn : nullptr);
        crf_filter<5>(lat[1], q1, true, lat[1].norm, fb, false, val0, val1, NT, Ci, CS, 0, 1, s, first ? lat[1].kn : nullptr);
        const bool last = it + 1 == iters;
        hipLaunchKernelGGL(crf_update_kernel, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, s, u, fg, fb, lat[0].kn, lat[1].kn, compat_g, compat_b,
                           q1, last ? mask : nullptr, last ? q_out : nullptr, NT, N, Ci, CS, (int)W, tw);
    }
    SS_LAUNCH_CHECK("dense_crf");
    return 0;
}
