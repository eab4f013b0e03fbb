// Row kernels behind the similarity matrices: InfoNCE cross-entropy rows (K13), retrieval first-match ranks (K16),
// and the fused AdamW step over the flat parameter buffer.
#include "common.h"

namespace {

__device__ __forceinline__ float block_reduce_sum(float v, float* sh) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += sh[i];
    return s;
}
__device__ __forceinline__ float block_reduce_max(float v, float* sh) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = -INFINITY;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s = fmaxf(s, sh[i]);
    return s;
}

// One block per local row i of sims[N1,N2] = feat1 . feat2_global^T  (mml_loss.py:73-77, 89-95, 350-376).
//   z = s / clamp(T, 1e-3, 0.5);  loss_i = (1-eps) * nll_i + eps * mean_j(-logp_ij);  weight w_i = 1 - ignore_i
//   total loss = (1/N1) sum_i w_i loss_i.   Writes (in place) dS_ij = dLoss/ds_ij, and per-row partials.
__global__ __launch_bounds__(256) void nce_rows_kernel(float* __restrict__ sims, const float* __restrict__ temperature,
                                                       const float* __restrict__ ignore, float* __restrict__ row_loss,
                                                       float* __restrict__ row_correct, float* __restrict__ row_tdot,
                                                       int N1, int N2, int target0, float smoothing, int write_grad) {
    // (the pair form - simseg_nce_pair - launches 2 N1 blocks over two stacked [N1, N2] matrices: row i of matrix i / N1)
    __shared__ float sh[8];
    __shared__ int shi[8];
    const int i = blockIdx.x, tid = threadIdx.x;
    float* row = sims + (long)i * N2;
    const float temp = fminf(fmaxf(temperature[0], 0.001f), 0.5f);
    const float inv_t = 1.0f / temp;
    const int il = i % N1;
    const int tgt = target0 + il;
    float mx = -INFINITY;
    int arg = 0x7fffffff;
    for (int j = tid; j < N2; j += 256) {
        const float z = row[j] * inv_t;
        if (z > mx) { mx = z; arg = j; }
    }
    const float bmx = block_reduce_max(mx, sh);
    // first index attaining the maximum
    int cand = (mx == bmx) ? arg : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
    __syncthreads();
    if ((tid & 63) == 0) shi[tid >> 6] = cand;
    __syncthreads();
    int best = 0x7fffffff;
    for (int w = 0; w < 4; ++w) best = min(best, shi[w]);
    float se = 0.f, sz = 0.f;
    for (int j = tid; j < N2; j += 256) {
        const float z = row[j] * inv_t;
        se += __expf(z - bmx);
        sz += z;
    }
    se = block_reduce_sum(se, sh);
    sz = block_reduce_sum(sz, sh);
    const float lse = bmx + __logf(se);
    const float zt = row[tgt] * inv_t;
    const float nll = lse - zt;
    const float smooth = lse - sz / N2;                 // -mean_j logp_ij
    const float w = ignore ? 1.0f - ignore[il] : 1.0f;
    const float loss = (1.0f - smoothing) * nll + smoothing * smooth;
    __syncthreads();
    float tdot = 0.f;
    if (write_grad) {
        const float c = w / N1;
        for (int j = tid; j < N2; j += 256) {
            const float s = row[j];
            const float pj = __expf(s * inv_t - lse);
            float dz = pj - smoothing / N2;
            if (j == tgt) dz -= (1.0f - smoothing);
            dz *= c;
            tdot += dz * s;
            row[j] = dz * inv_t;
        }
        tdot = block_reduce_sum(tdot, sh);
    }
    if (tid == 0) {
        row_loss[i] = loss * w;
        row_correct[i] = (best == tgt) ? 1.f : 0.f;
        row_tdot[i] = tdot;
    }
}

// out[0] = loss = mean_i row_loss ; out[1] = top-1 acc over kept rows ; out[2] = dLoss/dTemperature
__global__ __launch_bounds__(256) void nce_finalize_kernel(const float* __restrict__ row_loss, const float* __restrict__ row_correct,
                                                           const float* __restrict__ row_tdot, const float* __restrict__ ignore,
                                                           const float* __restrict__ temperature, float* __restrict__ out, int N1) {
    __shared__ float sh[8];
    float l = 0.f, c = 0.f, k = 0.f, t = 0.f;
    for (int i = threadIdx.x; i < N1; i += 256) {
        const bool keep = !ignore || ignore[i] < 1.0f;
        l += row_loss[i];
        t += row_tdot[i];
        if (keep) { c += row_correct[i]; k += 1.f; }
    }
    l = block_reduce_sum(l, sh); c = block_reduce_sum(c, sh); k = block_reduce_sum(k, sh); t = block_reduce_sum(t, sh);
    if (threadIdx.x == 0) {
        const float T = temperature[0];
        const float temp = fminf(fmaxf(T, 0.001f), 0.5f);
        out[0] = l / N1;
        out[1] = c / k;
        // z = s / temp  ->  dL/dtemp = -(1/temp) * sum_ij dz_ij * s_ij / temp ... with tdot = sum dz*s:  -tdot / temp^2
        out[2] = (T >= 0.001f && T <= 0.5f) ? -t / (temp * temp) : 0.f;
    }
}

// Both directions of the CLIP loss at once (pipelines/clip.py:129-140: 0.5 (i2t + t2i)): rows [0, N1) are the image -> text matrix, rows
// [N1, 2 N1) the text -> image one.  out = {loss, i2t acc, t2i acc, dLoss/dTemperature}.
__global__ __launch_bounds__(256) void nce_finalize_pair_kernel(const float* __restrict__ row_loss, const float* __restrict__ row_correct,
                                                                const float* __restrict__ row_tdot, const float* __restrict__ temperature,
                                                                float* __restrict__ out, int N1) {
    __shared__ float sh[8];
    float l = 0.f, c0 = 0.f, c1 = 0.f, t = 0.f;
    for (int i = threadIdx.x; i < 2 * N1; i += 256) {
        l += row_loss[i];
        t += row_tdot[i];
        if (i < N1) c0 += row_correct[i]; else c1 += row_correct[i];
    }
    l = block_reduce_sum(l, sh); c0 = block_reduce_sum(c0, sh); c1 = block_reduce_sum(c1, sh); t = block_reduce_sum(t, sh);
    if (threadIdx.x == 0) {
        const float T = temperature[0];
        const float temp = fminf(fmaxf(T, 0.001f), 0.5f);
        out[0] = 0.5f * l / N1;
        out[1] = c0 / N1;
        out[2] = c1 / N1;
        out[3] = (T >= 0.001f && T <= 0.5f) ? -0.5f * t / (temp * temp) : 0.f;
    }
}

// The loss head's backward preparation in ONE launch: up to six fp32 matrices are transposed (32 x 32 LDS tiles), the ones flagged `scale`
// are multiplied by alpha * scalar[0] on the way - in the transposed copy AND in place - and y0[0] = scalar[0] * x0[0].  (The fp32
// GEMM kernel contracts row . row: its transposed operands are made here, together with the upstream-gradient scale the round-3 loss head
// applied in separate passes.)
struct TransposeJob { const float* in; float* out; int R, C, scale, tiles_c, tile0; };
struct TransposeJobs { TransposeJob j[6]; int n; const float* scalar; float alpha; const float* x0; float* y0; };
__global__ __launch_bounds__(256) void transpose_multi_kernel(TransposeJobs J) {
    __shared__ float t[32][33];
    int k = 0;
    while (k + 1 < J.n && (int)blockIdx.x >= J.j[k + 1].tile0) ++k;
    const TransposeJob jb = J.j[k];
    const int tile = blockIdx.x - jb.tile0;
    const int c0 = (tile % jb.tiles_c) * 32, r0 = (tile / jb.tiles_c) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float f = jb.scale ? J.alpha * J.scalar[0] : 1.0f;
    float* inw = const_cast<float*>(jb.in);
    for (int j = ty; j < 32; j += 8)
        if (r0 + j < jb.R && c0 + tx < jb.C) {
            const float v = jb.in[(long)(r0 + j) * jb.C + c0 + tx] * f;
            t[j][tx] = v;
            if (jb.scale) inw[(long)(r0 + j) * jb.C + c0 + tx] = v;
        }
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (c0 + j < jb.C && r0 + tx < jb.R) jb.out[(long)(c0 + j) * jb.R + r0 + tx] = t[tx][j];
    if (blockIdx.x == 0 && threadIdx.x == 0 && J.y0) J.y0[0] = J.scalar[0] * J.x0[0];
}

// y[r,:] = x[r,:] * (one_minus ? 1 - s[r] : s[r]) * alpha
__global__ void scale_rows_kernel(const float* __restrict__ x, const float* __restrict__ s, float* __restrict__ y, long rows,
                                  int D, int one_minus, float alpha) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < rows * D; i += (long)gridDim.x * blockDim.x) {
        const float f = s ? (one_minus ? 1.0f - s[i / D] : s[i / D]) : 1.0f;
        y[i] = x[i] * f * alpha;
    }
}

// y[i] = x[i] * scalar[0] * alpha  (upstream scalar gradient applied without a host round trip)
__global__ void scale_by_scalar_kernel(const float* __restrict__ x, const float* __restrict__ scalar, float* __restrict__ y, long n,
                                       float alpha) {
    const float f = scalar[0] * alpha;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = x[i] * f;
}

// Retrieval (hooks/utils.py:36-42, 64-66): for left row i, best = max_j{sim_ij : gid match}; rank = #{j : sim_ij > best}.
__global__ __launch_bounds__(256) void retrieval_rank_kernel(const float* __restrict__ sim, const long* __restrict__ lgid,
                                                             const long* __restrict__ rgid, int* __restrict__ has,
                                                             int* __restrict__ rank, int N, long ld) {
    __shared__ float sh[8];
    const long i = blockIdx.x;
    const float* row = sim + i * ld;
    const long g = lgid[i];
    float best = -INFINITY;
    for (int j = threadIdx.x; j < N; j += 256)
        if (rgid[j] == g) best = fmaxf(best, row[j]);
    best = block_reduce_max(best, sh);
    float cnt = 0.f;
    for (int j = threadIdx.x; j < N; j += 256) cnt += (row[j] > best) ? 1.f : 0.f;
    cnt = block_reduce_sum(cnt, sh);
    if (threadIdx.x == 0) {
        has[i] = best > -INFINITY ? 1 : 0;
        rank[i] = (int)cnt;
    }
}

// The reverse direction from the SAME similarity matrix (columns retrieve rows): for column j, best = max_i{sim_ij : gid match},
// rank = #{i : sim_ij > best}.  Two row-major passes over the matrix (thread per column, a block covers 256 columns x a band of rows;
// per-column partial results meet in one atomic per block) instead of a second M x N GEMM on swapped operands.
__device__ __forceinline__ int ord_f32(float f) { const int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
__device__ __forceinline__ float unord_f32(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }
__global__ __launch_bounds__(256) void retrieval_cols_init_kernel(int* __restrict__ best, int* __restrict__ rank, int N) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j < N) { best[j] = ord_f32(-INFINITY); rank[j] = 0; }
}
template <int PASS>
__global__ __launch_bounds__(256) void retrieval_cols_kernel(const float* __restrict__ sim, const long* __restrict__ rgid_rows,
                                                             const long* __restrict__ cgid, int* __restrict__ best, int* __restrict__ has,
                                                             int* __restrict__ rank, int M, int N, long ld, int band) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int r0 = blockIdx.y * band, r1 = min(M, r0 + band);
    if (j >= N) return;
    const float* col = sim + j;
    if (PASS == 0) {
        const long g = cgid[j];
        float b = -INFINITY;
        for (int r = r0; r < r1; ++r)
            if (rgid_rows[r] == g) b = fmaxf(b, col[(long)r * ld]);
        if (b > -INFINITY) atomicMax(best + j, ord_f32(b));
    } else {
        const float b = unord_f32(best[j]);
        int cnt = 0;
        for (int r = r0; r < r1; ++r) cnt += col[(long)r * ld] > b ? 1 : 0;
        if (cnt) atomicAdd(rank + j, cnt);
        if (blockIdx.y == 0) has[j] = b > -INFINITY ? 1 : 0;
    }
}

// counts[0] = #rows with a match, counts[1..nb] = #rows with a match and rank < bound_b
__global__ __launch_bounds__(256) void recall_count_kernel(const int* __restrict__ has, const int* __restrict__ rank, long M,
                                                           int b0, int b1, int b2, int* __restrict__ counts) {
    int c[4] = {0, 0, 0, 0};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (long)gridDim.x * blockDim.x) {
        if (has[i]) {
            c[0]++;
            c[1] += rank[i] < b0; c[2] += rank[i] < b1; c[3] += rank[i] < b2;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int v = c[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(counts + k, v);
    }
}

// torch.optim.AdamW step on a flat fp32 segment, optionally refreshing the bf16 compute copy of the weights.
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             bf16_t* __restrict__ p16, long n, float lr, float b1, float b2, float eps, float wd, float bc1,
                             float bc2_sqrt, float grad_scale) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gi = g[i] * grad_scale;
        float pi = p[i] * (1.0f - lr * wd);
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        pi -= (lr / bc1) * (mi / denom);
        p[i] = pi; m[i] = mi; v[i] = vi;
        if (p16) p16[i] = (bf16_t)pi;
    }
}

// Multi-tensor form: one launch for every parameter tensor.  table[t] = {p, g, m, v, p16, lr, weight_decay}: five device pointers and
// the tensor's own learning rate / weight decay (the reference builds one param group per parameter, tasks/clip/hooks/optimizer.py:18-36);
// sizes[t] elements; chunk c of the launch covers elements [chunk_off[c], chunk_off[c] + chunk) of tensor chunk_tid[c].  p16 (may be
// null) is the bf16 compute copy the next forward's GEMMs read: refreshed here, so no per-step cast kernels.
struct AdamTensors { float* p; const float* g; float* m; float* v; bf16_t* p16; float lr; float wd; };
static_assert(sizeof(AdamTensors) == 48, "table rows are six 8-byte words");
// AMP form (round 4): the loss scale, the overflow flag and the count of steps actually taken live on the DEVICE (amp.scale / amp.found_inf:
// torch.amp.GradScaler's tensors, handed over through its `optimizer.grad_scale` / `optimizer.found_inf` contract; amp.step_in / step_out:
// this optimizer's own two-slot counter).  The gradients are unscaled here (g / scale), and a step whose gradients held an inf / nan updates
// nothing - the decision never travels to the host, so the reference's fp16 iteration (clip_runner.py:226-230, core/hooks/optimizer.py:73-82)
// runs without the host read torch's scaler.step() makes for an optimizer that cannot skip by itself.  Bias corrections come from the device
// counter (skipped steps do not count, as in torch's own fused Adam).
struct AdamAmp { const float* scale; const float* found_inf; const float* step_in; float* step_out; };
__global__ __launch_bounds__(256) void adamw_multi_kernel(const AdamTensors* __restrict__ table, const long* __restrict__ sizes,
                                                          const int* __restrict__ chunk_tid, const long* __restrict__ chunk_off,
                                                          int chunk, float b1, float b2, float eps, float bc1, float bc2_sqrt,
                                                          float grad_scale, AdamAmp amp) {
    const int c = blockIdx.x;
    if (amp.step_in) {
        const bool skip = amp.found_inf && amp.found_inf[0] != 0.f;
        const float st = amp.step_in[0] + (skip ? 0.f : 1.f);
        if (c == 0 && threadIdx.x == 0) amp.step_out[0] = st;      // (the other slot: nobody reads it during this launch)
        if (skip) return;
        bc1 = 1.0f - powf(b1, st);
        bc2_sqrt = sqrtf(1.0f - powf(b2, st));
        if (amp.scale) grad_scale /= amp.scale[0];
    }
    const int t = chunk_tid[c];
    const AdamTensors T = table[t];
    const long lo = chunk_off[c];
    const long hi = min(sizes[t], lo + chunk);
    const float decay = 1.0f - T.lr * T.wd;
    const float step_size = T.lr / bc1;
    auto one = [&](float g_, float& pi, float& mi, float& vi) {
        const float gi = g_ * grad_scale;
        pi *= decay;
        mi = b1 * mi + (1.0f - b1) * gi;
        vi = b2 * vi + (1.0f - b2) * gi * gi;
        pi -= step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
    };
    // 16 bytes per lane and streaming (non-temporal) accesses when the chunk allows it: every array is touched exactly once per step
    // (30 bytes per parameter, 5.9 GB for ViT-B + BERT-base)
    typedef float f4 __attribute__((ext_vector_type(4)));
    const bool vec = (((uintptr_t)(T.g + lo) | (uintptr_t)(T.p + lo) | (uintptr_t)(T.m + lo) | (uintptr_t)(T.v + lo)) % 16 == 0) &&
                     (!T.p16 || (uintptr_t)(T.p16 + lo) % 8 == 0);
    long i0 = lo;
    if (vec) {
        const long n4 = (hi - lo) / 4;
        for (long q = threadIdx.x; q < n4; q += 256) {
            const long i = lo + 4 * q;
            const f4 g4 = __builtin_nontemporal_load(reinterpret_cast<const f4*>(T.g + i));
            f4 p4 = __builtin_nontemporal_load(reinterpret_cast<const f4*>(T.p + i));
            f4 m4 = __builtin_nontemporal_load(reinterpret_cast<const f4*>(T.m + i));
            f4 v4 = __builtin_nontemporal_load(reinterpret_cast<const f4*>(T.v + i));
#pragma unroll
            for (int e = 0; e < 4; ++e) { float pe = p4[e], me = m4[e], ve = v4[e]; one(g4[e], pe, me, ve); p4[e] = pe; m4[e] = me; v4[e] = ve; }
            __builtin_nontemporal_store(p4, reinterpret_cast<f4*>(T.p + i));
            __builtin_nontemporal_store(m4, reinterpret_cast<f4*>(T.m + i));
            __builtin_nontemporal_store(v4, reinterpret_cast<f4*>(T.v + i));
            if (T.p16) {
                bf16x4 o = {(bf16_t)p4[0], (bf16_t)p4[1], (bf16_t)p4[2], (bf16_t)p4[3]};
                *reinterpret_cast<bf16x4*>(T.p16 + i) = o;             // (read by the next step's GEMMs: default policy)
            }
        }
        i0 = lo + 4 * n4;
    }
    for (long i = i0 + threadIdx.x; i < hi; i += 256) {
        float pi = T.p[i], mi = T.m[i], vi = T.v[i];
        one(T.g[i], pi, mi, vi);
        T.p[i] = pi; T.m[i] = mi; T.v[i] = vi;
        if (T.p16) T.p16[i] = (bf16_t)pi;
    }
}

// found[0] = 1 if any gradient element of the table's tensors is inf / nan (read-only pass, 16 bytes per lane, one store per block that
// sees one): the overflow check of torch.amp.GradScaler (`_amp_foreach_non_finite_check_and_unscale_` with inverse scale 1: a read-modify-
// write pass over every gradient tensor) as ONE launch over the optimizer's own tensor table; the unscaling itself rides on the AdamW kernel.
__global__ __launch_bounds__(256) void grads_nonfinite_kernel(const AdamTensors* __restrict__ table, const long* __restrict__ sizes,
                                                              const int* __restrict__ chunk_tid, const long* __restrict__ chunk_off,
                                                              int chunk, float* __restrict__ found) {
    const int c = blockIdx.x;
    const int t = chunk_tid[c];
    const float* g = table[t].g;
    const long lo = chunk_off[c];
    const long hi = min(sizes[t], lo + chunk);
    unsigned bad = 0;
    long i0 = lo;
    if ((uintptr_t)(g + lo) % 16 == 0) {
        const long n4 = (hi - lo) / 4;
        for (long q = threadIdx.x; q < n4; q += 256) {
            const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(g + lo + 4 * q));
#pragma unroll
            for (int e = 0; e < 4; ++e) bad |= ((v[e] & 0x7f800000u) == 0x7f800000u) ? 1u : 0u;
        }
        i0 = lo + 4 * n4;
    }
    for (long i = i0 + threadIdx.x; i < hi; i += 256) bad |= ((__float_as_uint(g[i]) & 0x7f800000u) == 0x7f800000u) ? 1u : 0u;
    if (__any(bad != 0) && (threadIdx.x & 63) == 0) found[0] = 1.0f;
}

}  // namespace

#define STREAM ((hipStream_t)stream)

extern "C" int simseg_nce_rows(float* sims, const float* temperature, const float* ignore_mask, float* row_loss, float* row_correct,
                               float* row_tdot, float* out3, int64_t N1, int64_t N2, int64_t target0, float smoothing,
                               int write_grad, void* stream) {
    SS_CHECK(sims && temperature && row_loss && row_correct && row_tdot && out3, "nce_rows: null pointer");
    SS_CHECK(N1 > 0 && N2 > 0 && target0 >= 0 && target0 + N1 <= N2, "nce_rows: targets [%lld, %lld) outside [0, %lld)",
             (long long)target0, (long long)(target0 + N1), (long long)N2);
    hipLaunchKernelGGL(nce_rows_kernel, dim3((unsigned)N1), dim3(256), 0, STREAM, sims, temperature, ignore_mask, row_loss, row_correct,
                       row_tdot, (int)N1, (int)N2, (int)target0, smoothing, write_grad);
    hipLaunchKernelGGL(nce_finalize_kernel, dim3(1), dim3(256), 0, STREAM, row_loss, row_correct, row_tdot, ignore_mask, temperature,
                       out3, (int)N1);
    SS_LAUNCH_CHECK("nce_rows");
    return 0;
}

extern "C" int simseg_nce_pair(float* sims2, const float* temperature, float* row_scratch, float* out4, int64_t N1, int64_t N2, int64_t target0,
                               float smoothing, int write_grad, void* stream) {
    SS_CHECK(sims2 && temperature && row_scratch && out4, "nce_pair: null pointer");
    SS_CHECK(N1 > 0 && N2 > 0 && target0 >= 0 && target0 + N1 <= N2, "nce_pair: targets [%lld, %lld) outside [0, %lld)",
             (long long)target0, (long long)(target0 + N1), (long long)N2);
    float* rl = row_scratch; float* rc = rl + 2 * N1; float* rt = rc + 2 * N1;
    hipLaunchKernelGGL(nce_rows_kernel, dim3((unsigned)(2 * N1)), dim3(256), 0, STREAM, sims2, temperature, (const float*)nullptr, rl, rc, rt, (int)N1,
                       (int)N2, (int)target0, smoothing, write_grad);
    hipLaunchKernelGGL(nce_finalize_pair_kernel, dim3(1), dim3(256), 0, STREAM, rl, rc, rt, temperature, out4, (int)N1);
    SS_LAUNCH_CHECK("nce_pair");
    return 0;
}

extern "C" int simseg_transpose_multi(const void* const* in, void* const* out, const int64_t* rows, const int64_t* cols, const int32_t* scale,
                                      int64_t n, const float* scalar, float alpha, const float* x0, float* y0, void* stream) {
    SS_CHECK(in && out && rows && cols && scale && n >= 1 && n <= 6, "transpose_multi: 1..6 jobs");
    TransposeJobs J;
    memset(&J, 0, sizeof(J));
    int tile0 = 0, any = 0;
    for (int k = 0; k < (int)n; ++k) {
        SS_CHECK(in[k] && out[k] && rows[k] > 0 && cols[k] > 0, "transpose_multi: job %d is empty", k);
        TransposeJob& j = J.j[k];
        j.in = (const float*)in[k]; j.out = (float*)out[k]; j.R = (int)rows[k]; j.C = (int)cols[k]; j.scale = scale[k];
        j.tiles_c = (j.C + 31) / 32; j.tile0 = tile0;
        tile0 += j.tiles_c * ((j.R + 31) / 32);
        any |= scale[k];
    }
    SS_CHECK(!(any || y0) || scalar, "transpose_multi: a scaled job needs the scalar");
    SS_CHECK(!y0 || x0, "transpose_multi: y0 needs x0");
    J.n = (int)n; J.scalar = scalar; J.alpha = alpha; J.x0 = x0; J.y0 = y0;
    hipLaunchKernelGGL(transpose_multi_kernel, dim3((unsigned)tile0), dim3(256), 0, STREAM, J);
    SS_LAUNCH_CHECK("transpose_multi");
    return 0;
}

extern "C" int simseg_scale_rows(const float* x, const float* s, float* y, int64_t rows, int64_t D, int one_minus, float alpha,
                                 void* stream) {
    SS_CHECK(x && y, "scale_rows: null pointer");
    if (rows * D <= 0) return 0;
    long blocks = (rows * D + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, STREAM, x, s, y, (long)rows, (int)D, one_minus, alpha);
    SS_LAUNCH_CHECK("scale_rows");
    return 0;
}

extern "C" int simseg_scale_by_scalar(const float* x, const float* scalar, float* y, int64_t n, float alpha, void* stream) {
    SS_CHECK(x && scalar && y, "scale_by_scalar: null pointer");
    if (n <= 0) return 0;
    long blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(scale_by_scalar_kernel, dim3((unsigned)blocks), dim3(256), 0, STREAM, x, scalar, y, (long)n, alpha);
    SS_LAUNCH_CHECK("scale_by_scalar");
    return 0;
}

extern "C" int simseg_retrieval_rank(const float* sim, const int64_t* left_gid, const int64_t* right_gid, int32_t* has_match,
                                     int32_t* rank, int64_t M, int64_t N, int64_t ld, void* stream) {
    SS_CHECK(sim && left_gid && right_gid && has_match && rank, "retrieval_rank: null pointer");
    if (M <= 0) return 0;
    hipLaunchKernelGGL(retrieval_rank_kernel, dim3((unsigned)M), dim3(256), 0, STREAM, sim, (const long*)left_gid, (const long*)right_gid,
                       has_match, rank, (int)N, (long)ld);
    SS_LAUNCH_CHECK("retrieval_rank");
    return 0;
}

extern "C" int simseg_retrieval_rank_cols(const float* sim, const int64_t* row_gid, const int64_t* col_gid, int32_t* has_match,
                                          int32_t* rank, int32_t* scratch, int64_t M, int64_t N, int64_t ld, void* stream) {
    SS_CHECK(sim && row_gid && col_gid && has_match && rank && scratch, "retrieval_rank_cols: null pointer");
    if (N <= 0 || M <= 0) return 0;
    const int band = 128;
    dim3 grid((unsigned)((N + 255) / 256), (unsigned)((M + band - 1) / band));
    hipLaunchKernelGGL(retrieval_cols_init_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, STREAM, scratch, rank, (int)N);
    hipLaunchKernelGGL(retrieval_cols_kernel<0>, grid, dim3(256), 0, STREAM, sim, (const long*)row_gid, (const long*)col_gid, scratch, has_match, rank,
                       (int)M, (int)N, (long)ld, band);
    hipLaunchKernelGGL(retrieval_cols_kernel<1>, grid, dim3(256), 0, STREAM, sim, (const long*)row_gid, (const long*)col_gid, scratch, has_match, rank,
                       (int)M, (int)N, (long)ld, band);
    SS_LAUNCH_CHECK("retrieval_rank_cols");
    return 0;
}

extern "C" int simseg_recall_counts(const int32_t* has_match, const int32_t* rank, int64_t M, int b0, int b1, int b2,
                                    int32_t* counts4, void* stream) {
    SS_CHECK(has_match && rank && counts4, "recall_counts: null pointer");
    (void)hipMemsetAsync(counts4, 0, 4 * sizeof(int), STREAM);
    long blocks = (M + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(recall_count_kernel, dim3((unsigned)blocks), dim3(256), 0, STREAM, has_match, rank, (long)M, b0, b1, b2, counts4);
    SS_LAUNCH_CHECK("recall_counts");
    return 0;
}

extern "C" int simseg_adamw_multi_step(const void* table, const int64_t* sizes, const int32_t* chunk_tid, const int64_t* chunk_off,
                                       int64_t n_chunks, int chunk, float beta1, float beta2, float eps, int64_t step,
                                       float grad_scale, void* stream) {
    SS_HALF_FWD(simseg_adamw_multi_step, table, sizes, chunk_tid, chunk_off, n_chunks, chunk, beta1, beta2, eps, step, grad_scale, stream);
    SS_CHECK(table && sizes && chunk_tid && chunk_off, "adamw_multi_step: null pointer");
    SS_CHECK(step >= 1 && chunk > 0, "adamw_multi_step: bad step/chunk");
    if (n_chunks <= 0) return 0;
    const float bc1 = 1.0f - powf(beta1, (float)step);
    const float bc2 = 1.0f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adamw_multi_kernel, dim3((unsigned)n_chunks), dim3(256), 0, STREAM, (const AdamTensors*)table, (const long*)sizes,
                       (const int*)chunk_tid, (const long*)chunk_off, chunk, beta1, beta2, eps, bc1, sqrtf(bc2), grad_scale, AdamAmp{});
    SS_LAUNCH_CHECK("adamw_multi_step");
    return 0;
}

extern "C" int simseg_adamw_multi_step_amp(const void* table, const int64_t* sizes, const int32_t* chunk_tid, const int64_t* chunk_off,
                                           int64_t n_chunks, int chunk, float beta1, float beta2, float eps, float grad_scale,
                                           const float* loss_scale, const float* found_inf, const float* step_in, float* step_out, void* stream) {
    SS_HALF_FWD(simseg_adamw_multi_step_amp, table, sizes, chunk_tid, chunk_off, n_chunks, chunk, beta1, beta2, eps, grad_scale, loss_scale,
                found_inf, step_in, step_out, stream);
    SS_CHECK(table && sizes && chunk_tid && chunk_off && step_in && step_out && step_in != step_out, "adamw_multi_step_amp: null / aliased pointer");
    SS_CHECK(chunk > 0, "adamw_multi_step_amp: bad chunk");
    if (n_chunks <= 0) return 0;
    AdamAmp amp{loss_scale, found_inf, step_in, step_out};
    hipLaunchKernelGGL(adamw_multi_kernel, dim3((unsigned)n_chunks), dim3(256), 0, STREAM, (const AdamTensors*)table, (const long*)sizes,
                       (const int*)chunk_tid, (const long*)chunk_off, chunk, beta1, beta2, eps, 1.0f, 1.0f, grad_scale, amp);
    SS_LAUNCH_CHECK("adamw_multi_step_amp");
    return 0;
}

extern "C" int simseg_grads_nonfinite(const void* table, const int64_t* sizes, const int32_t* chunk_tid, const int64_t* chunk_off,
                                      int64_t n_chunks, int chunk, float* found_inf, void* stream) {
    SS_HALF_FWD(simseg_grads_nonfinite, table, sizes, chunk_tid, chunk_off, n_chunks, chunk, found_inf, stream);
    SS_CHECK(table && sizes && chunk_tid && chunk_off && found_inf, "grads_nonfinite: null pointer");
    SS_CHECK(chunk > 0, "grads_nonfinite: bad chunk");
    if (n_chunks <= 0) return 0;
    hipLaunchKernelGGL(grads_nonfinite_kernel, dim3((unsigned)n_chunks), dim3(256), 0, STREAM, (const AdamTensors*)table, (const long*)sizes,
                       (const int*)chunk_tid, (const long*)chunk_off, chunk, found_inf);
    SS_LAUNCH_CHECK("grads_nonfinite");
    return 0;
}

extern "C" int simseg_adamw_step(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr, float beta1,
                                 float beta2, float eps, float weight_decay, int64_t step, float grad_scale, void* stream) {
    SS_HALF_FWD(simseg_adamw_step, p, g, m, v, p_bf16, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, stream);
    SS_CHECK(p && g && m && v, "adamw_step: null pointer");
    SS_CHECK(step >= 1, "adamw_step: step counts from 1");
    if (n <= 0) return 0;
    const float bc1 = 1.0f - powf(beta1, (float)step);
    const float bc2 = 1.0f - powf(beta2, (float)step);
    long blocks = (n + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, STREAM, p, g, m, v, (bf16_t*)p_bf16, (long)n, lr, beta1, beta2, eps,
                       weight_decay, bc1, sqrtf(bc2), grad_scale);
    SS_LAUNCH_CHECK("adamw_step");
    return 0;
}
