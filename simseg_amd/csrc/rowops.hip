// HBM-bound row kernels of the encoder path: LayerNorm fwd/bwd (K3), patch gather (K1), [cls]+pos rows (K2),
// BERT embedding gather/scatter (K9), LoDA top-k pooling + L2norm fwd/bwd (K11, K12), row norms, column sums,
// casts and transposes.  Every kernel moves 16 bytes per lane per access and reduces rows with wave shuffles.
#include "common.h"
#include <atomic>
#include <stdlib.h>

namespace {

constexpr int LN_MAXC = 8;   // float4 chunks per lane -> D <= 2048

template <typename TO> __device__ __forceinline__ void store4(TO* p, const float (&v)[4]);
template <> __device__ __forceinline__ void store4<float>(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, const float (&v)[4]) {
    bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    *reinterpret_cast<bf16x4*>(p) = o;
}
template <typename TI> __device__ __forceinline__ void load4(const TI* p, float (&v)[4]);
template <> __device__ __forceinline__ void load4<float>(const float* p, float (&v)[4]) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <> __device__ __forceinline__ void load4<bf16_t>(const bf16_t* p, float (&v)[4]) {
    const bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
    v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
}

// streaming variants: data that is read once / not re-read before it has left every cache anyway (the fp32 residual stream: 310 MB per
// tensor at the training shape, more than the 256 MB memory-side cache)
typedef float f32x4_nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void load4_nt(const float* p, float (&v)[4]) {
    const f32x4_nt t = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(p));
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
}
__device__ __forceinline__ void store4_nt(float* p, const float (&v)[4]) {
    const f32x4_nt t = {v[0], v[1], v[2], v[3]};
    __builtin_nontemporal_store(t, reinterpret_cast<f32x4_nt*>(p));
}

// ---------------------------------------------------------------------------------------------
// LayerNorm forward: one wave per row.  x fp32 (residual stream) -> y (TO) [+ optional bf16 copy y2]
// timm Block.norm1/norm2/norm (eps 1e-6), HF BertSelfOutput/BertOutput/BertEmbeddings LayerNorm (eps 1e-12)
// ---------------------------------------------------------------------------------------------
template <typename TO>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, TO* __restrict__ y,
                                                     bf16_t* __restrict__ y2, float* __restrict__ mean_out,
                                                     float* __restrict__ rstd_out, int rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));     // wave-uniform: scalar row bases
    if (row >= rows) return;
    const int nch = D >> 2;
    const float* xr = x + (long)row * D;
    float v[LN_MAXC][4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            load4_nt(xr + c * 4, v[i]);
            s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
    }
    const float mean = wave_sum_uniform(s) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mean; q += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum_uniform(q) / D + eps);
    if (lane == 0) {
        if (mean_out) mean_out[row] = mean;
        if (rstd_out) rstd_out[row] = rstd;
    }
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            float g[4], b[4], o[4];
            load4<float>(gamma + c * 4, g);
            load4<float>(beta + c * 4, b);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mean) * rstd * g[j] + b[j];
            store4<TO>(y + (long)row * D + c * 4, o);
            if (y2) store4<bf16_t>(y2 + (long)row * D + c * 4, o);
        }
    }
}

// Narrow rows (D <= 512: ViT-S, the test towers): TWO rows per wave, 32 lanes each (round 5).  With one row per wave a 384-wide row
// is 96 chunks on 64 lanes - half the lanes idle on the second load - and the kernel ran at 3.9 TB/s against 5.8 at D = 768.
// Row sums: the DPP network leaves every lane of rows 1 / 3 with its half-wave's total; two readlanes hand them to both halves.
__device__ __forceinline__ float half_wave_sum(float v, int half) {
#define SS_DPP_ADD(ctrl, rmask) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rmask, 0xf, true))
    SS_DPP_ADD(0xB1, 0xf);      // quad_perm [1,0,3,2]
    SS_DPP_ADD(0x4E, 0xf);      // quad_perm [2,3,0,1]
    SS_DPP_ADD(0x141, 0xf);     // row_half_mirror
    SS_DPP_ADD(0x140, 0xf);     // row_mirror
    SS_DPP_ADD(0x142, 0xa);     // row_bcast15 -> rows 1, 3
#undef SS_DPP_ADD
    const float lo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31));
    const float hi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
    return half ? hi : lo;
}
template <typename TO>
__global__ __launch_bounds__(256) void ln_fwd2_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, TO* __restrict__ y,
                                                      bf16_t* __restrict__ y2, float* __restrict__ mean_out,
                                                      float* __restrict__ rstd_out, int rows, int D, float eps) {
    constexpr int MAXC2 = 4;                 // 16-byte chunks per lane: D <= 512
    const int lane = threadIdx.x & 63, half = lane >> 5, l32 = lane & 31;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;
    if (row0 >= rows) return;                // (wave-uniform)
    const bool valid = row0 + half < rows;
    const int row = valid ? row0 + half : row0;         // the odd tail: the idle half re-reads its neighbour's row and stores nothing
    const int nch = D >> 2;
    const float* xr = x + (long)row * D;
    float v[MAXC2][4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC2; ++i) {
        const int c = l32 + 32 * i;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[i][j] = 0.f;
        if (c < nch) {
            load4_nt(xr + c * 4, v[i]);
            s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
    }
    const float mean = half_wave_sum(s, half) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC2; ++i) {
        const int c = l32 + 32 * i;
        if (c < nch) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mean; q += d * d; }
        }
    }
    const float rstd = rsqrtf(half_wave_sum(q, half) / D + eps);
    if (l32 == 0 && valid) {
        if (mean_out) mean_out[row] = mean;
        if (rstd_out) rstd_out[row] = rstd;
    }
#pragma unroll
    for (int i = 0; i < MAXC2; ++i) {
        const int c = l32 + 32 * i;
        if (c < nch && valid) {
            float g[4], b[4], o[4];
            load4<float>(gamma + c * 4, g);
            load4<float>(beta + c * 4, b);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mean) * rstd * g[j] + b[j];
            store4<TO>(y + (long)row * D + c * 4, o);
            if (y2) store4<bf16_t>(y2 + (long)row * D + c * 4, o);
        }
    }
}

// LayerNorm backward.  dy = dy16 (bf16, optional) + dy32 (fp32, optional);  dx = ln_bwd(dy) + dres (optional)
// Writes dx32 (fp32 residual-gradient stream) and dx16 (bf16 copy fed to the next dgrad/wgrad GEMMs, optionally with
// the forward dropout mask of the producing dense layer re-applied); dgamma/dbeta and the column sums of dx16 (= the
// bias gradient of that dense layer) are accumulated with one atomic per column per block.
//
// Round 5: the per-column parameters (gamma, and with y16 1/gamma and beta) live in LDS instead of registers (round 4's y16 form held them
// next to the three column accumulators: 148 VGPRs at D = 768, 3 waves per SIMD, 768 of 1024 blocks resident), the row index is
// wave-uniform (scalar row bases), and the host sizes the grid from hipOccupancyMaxActiveBlocksPerMultiprocessor: one resident round
// whatever the register count.  The 16-bit ViT step's form (dy16 + y16 + dres16 -> dx16) runs ln_bwd16_kernel below.
template <int MAXC>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16_t* __restrict__ dy16, const float* __restrict__ dy32,
                                                     const float* __restrict__ dres, const bf16_t* __restrict__ dres16,
                                                     const float* __restrict__ x, const bf16_t* __restrict__ y16, const float* __restrict__ beta,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, float* __restrict__ dx32,
                                                     bf16_t* __restrict__ dx16, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, float* __restrict__ dxsum, float* __restrict__ partials,
                                                     int rows, int D,
                                                     unsigned long long drop_seed, unsigned int drop_thresh, float drop_scale) {
    __shared__ float red[3][1024];   // [dgamma|dbeta|dxsum][wave * 256 + lane * 4 + j]
    extern __shared__ float par[];   // [gamma | 1/gamma | beta][D]   (the last two only with y16)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nch = D >> 2;
    float ag[MAXC][4], ab[MAXC][4], as[MAXC][4];
    // y16 (round 4): the normalised value xhat = (x - mean) * rstd is taken from the layer's saved 16-bit OUTPUT y = xhat * gamma + beta instead of
    // its fp32 input - half the bytes of the kernel's largest read - for every 4-channel chunk whose gains allow it (|gamma| >= 0.05 and
    // |beta| <= 4 |gamma|: the 16-bit rounding of y then perturbs xhat by <= 1 %); the other chunks read x as before, exactly.
    unsigned fromy = 0;              // bit i: chunk lane + 64 i takes xhat from y16
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { ag[i][j] = 0.f; ab[i][j] = 0.f; as[i][j] = 0.f; }
        const int c = lane + 64 * i;
        if (c < nch) {
            float gm[4];
            load4<float>(gamma + c * 4, gm);
            if (wave == 0) store4<float>(par + c * 4, gm);
            if (y16) {
                float bt[4], ig[4];
                load4<float>(beta + c * 4, bt);
                bool ok = true;
#pragma unroll
                for (int j = 0; j < 4; ++j) { ok = ok && fabsf(gm[j]) >= 0.05f && fabsf(bt[j]) <= 4.f * fabsf(gm[j]); ig[j] = 1.0f / gm[j]; }
                if (ok) fromy |= 1u << i;
                if (wave == 0) { store4<float>(par + D + c * 4, ig); store4<float>(par + 2 * D + c * 4, bt); }
            }
        }
    }
    __syncthreads();
    // (row index made wave-uniform for the compiler: row bases are then scalar registers and a lane's address is one 32-bit offset plus an
    //  immediate per chunk, instead of a 64-bit VGPR pair per pointer and chunk - 119 -> ~90 registers at D = 768)
    for (int row = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave); row < rows; row += gridDim.x * 4) {
        const float mu = mean[row], rs = rstd[row];
        const long ro = (long)row * D;
        float g[MAXC][4], xh[MAXC][4];
        float rr[MAXC][4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + 64 * i;
#pragma unroll
            for (int j = 0; j < 4; ++j) rr[i][j] = 0.f;
            if (c < nch) {
                const int o = c * 4;
                float a[4] = {0.f, 0.f, 0.f, 0.f}, b[4], xv[4], gm[4];
                if (dy16) {
                    const bf16x4 t = __builtin_nontemporal_load(reinterpret_cast<const bf16x4*>(dy16 + ro + o));
                    a[0] = (float)t[0]; a[1] = (float)t[1]; a[2] = (float)t[2]; a[3] = (float)t[3];
                }
                if (dy32) { load4<float>(dy32 + ro + o, b); for (int j = 0; j < 4; ++j) a[j] += b[j]; }
                if ((fromy >> i) & 1u) {
                    const bf16x4 t = __builtin_nontemporal_load(reinterpret_cast<const bf16x4*>(y16 + ro + o));
                    float ig[4], bt[4];
                    load4<float>(par + D + c * 4, ig);
                    load4<float>(par + 2 * D + c * 4, bt);
#pragma unroll
                    for (int j = 0; j < 4; ++j) xv[j] = ((float)t[j] - bt[j]) * ig[j];       // = xhat
                } else {
                    load4_nt(x + ro + o, xv);
#pragma unroll
                    for (int j = 0; j < 4; ++j) xv[j] = (xv[j] - mu) * rs;
                }
                if (dres) load4_nt(dres + ro + o, rr[i]);       // requested with the rest of the row: one memory round trip per row
                if (dres16) {                              // the residual gradient as the previous kernel's 16-bit copy (round 4: half the bytes)
                    const bf16x4 t = __builtin_nontemporal_load(reinterpret_cast<const bf16x4*>(dres16 + ro + o));
                    rr[i][0] += (float)t[0]; rr[i][1] += (float)t[1]; rr[i][2] += (float)t[2]; rr[i][3] += (float)t[3];
                }
                load4<float>(par + c * 4, gm);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    xh[i][j] = xv[j];
                    ag[i][j] += a[j] * xh[i][j];
                    ab[i][j] += a[j];
                    g[i][j] = a[j] * gm[j];
                    s1 += g[i][j];
                    s2 += g[i][j] * xh[i][j];
                }
            }
        }
        s1 = wave_sum_uniform(s1) / D;
        s2 = wave_sum_uniform(s2) / D;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < nch) {
                const int o = c * 4;
                float out[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) out[j] = rs * (g[i][j] - s1 - xh[i][j] * s2) + rr[i][j];
                if (dx32) store4_nt(dx32 + ro + o, out);
                if (drop_thresh) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) out[j] = dropout_keep(drop_seed, (unsigned long long)(ro + o) + j, drop_thresh) ? out[j] * drop_scale : 0.f;
                }
                if (dx16) store4<bf16_t>(dx16 + ro + o, out);
#pragma unroll
                for (int j = 0; j < 4; ++j) as[i][j] += out[j];
            }
        }
    }
    // cross-wave reduction of the column partials, then one atomic per column per block
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        __syncthreads();
        if (c < nch) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                red[0][wave * 256 + lane * 4 + j] = ag[i][j];
                red[1][wave * 256 + lane * 4 + j] = ab[i][j];
                red[2][wave * 256 + lane * 4 + j] = as[i][j];
            }
        }
        __syncthreads();
        if (wave == 0 && c < nch) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float sg = 0.f, sb = 0.f, ss = 0.f;
                for (int w = 0; w < 4; ++w) { sg += red[0][w * 256 + lane * 4 + j]; sb += red[1][w * 256 + lane * 4 + j]; ss += red[2][w * 256 + lane * 4 + j]; }
                if (partials) {      // [block][3][D] plain stores; ln_bwd_reduce_kernel folds the blocks (1 atomic per column in total)
                    float* pb = partials + (long)blockIdx.x * 3 * D + c * 4 + j;
                    pb[0] = sg; pb[D] = sb; pb[2 * D] = ss;
                } else {
                    atomicAdd(dgamma + c * 4 + j, sg);
                    atomicAdd(dbeta + c * 4 + j, sb);
                    if (dxsum) atomicAdd(dxsum + c * 4 + j, ss);
                }
            }
        }
    }
}

// The 16-bit ViT step's LayerNorm backward (round 5): dy16 + y16 + dres16 -> dx16, nothing fp32 in or out (8 bytes per element).
//   * straight-line row loop: every load of a row is issued before the first use (the generic kernel above waits per 4-channel chunk -
//     the compiler closes each `if (c < nch)` / `if (fromy)` region with s_waitcnt vmcnt(0), three loads in flight per wave); FULL = the
//     row is exactly MAXC * 256 channels, no lane predicate at all (D = 768, 512, 256, 1024), otherwise addresses are clamped and values
//     masked - still no branch around a load;
//   * the next row's dy / y loads are issued as soon as this row's raw values are converted: in flight during the two wave reductions,
//     the second pass and the stores; the residual gradient (second pass only) is requested at the top of the row;
//   * row index wave-uniform (scalar row bases), row sums through the DPP network into scalar registers (wave_sum_uniform: the
//     ds_bpermute butterfly cost 12 serialized LDS-crossbar round trips per row and 6 index registers);
//   * per-column parameters in LDS as in the generic kernel, re-read per row (sched_barriers keep the compiler from hoisting all nine
//     b128 reads - 36 registers - or double-buffering the prefetch in registers).
// Chunks whose gain fails the y16 test of the generic kernel (`okmask`; decided per launch, uniformly, `needx`) take xhat from the fp32
// input x, requested at the top of the row - same arithmetic as the generic kernel chunk by chunk.
template <int MAXC, bool FULL>
__global__ __launch_bounds__(256) void ln_bwd16_kernel(const bf16_t* __restrict__ dy16, const bf16_t* __restrict__ dres16,
                                                       const float* __restrict__ x, const bf16_t* __restrict__ y16, const float* __restrict__ beta,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       const float* __restrict__ gamma, bf16_t* __restrict__ dx16, float* __restrict__ dgamma,
                                                       float* __restrict__ dbeta, float* __restrict__ dxsum, float* __restrict__ partials,
                                                       int rows, int D,
                                                       unsigned long long drop_seed, unsigned int drop_thresh, float drop_scale) {
    __shared__ float red[3][1024];   // [dgamma|dbeta|dxsum][wave * 256 + lane * 4 + j]
    extern __shared__ float par[];   // [gamma | 1/gamma | beta][D]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nch = D >> 2;
    float ag[MAXC][4], ab[MAXC][4], as[MAXC][4];
    unsigned okmask = 0;             // bit i: chunk lane + 64 i takes xhat from y16
    int off[MAXC];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { ag[i][j] = 0.f; ab[i][j] = 0.f; as[i][j] = 0.f; }
        const int c = lane + 64 * i;
        off[i] = (FULL || c < nch) ? c * 4 : 0;
        if (FULL || c < nch) {
            float gm[4], bt[4], ig[4];
            load4<float>(gamma + c * 4, gm);
            load4<float>(beta + c * 4, bt);
            bool ok = true;
#pragma unroll
            for (int j = 0; j < 4; ++j) { ok = ok && fabsf(gm[j]) >= 0.05f && fabsf(bt[j]) <= 4.f * fabsf(gm[j]); ig[j] = 1.0f / gm[j]; }
            if (ok) okmask |= 1u << i;
            if (wave == 0) { store4<float>(par + c * 4, gm); store4<float>(par + D + c * 4, ig); store4<float>(par + 2 * D + c * 4, bt); }
        } else {
            okmask |= 1u << i;
        }
    }
    const bool needx = !__all(okmask == (1u << MAXC) - 1u);      // the same answer in every wave of the grid (each wave sees all of gamma / beta)
    __syncthreads();
    const int stride = gridDim.x * 4;
    int row = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
    bf16x4 rdy[MAXC], ry[MAXC];
#define LN16_ISSUE(R)                                                                                             \
    do {                                                                                                          \
        const long ro_ = (long)(R) * D;                                                                           \
        _Pragma("unroll") for (int i = 0; i < MAXC; ++i) {                                                        \
            rdy[i] = __builtin_nontemporal_load(reinterpret_cast<const bf16x4*>(dy16 + ro_ + off[i]));            \
            ry[i] = __builtin_nontemporal_load(reinterpret_cast<const bf16x4*>(y16 + ro_ + off[i]));              \
        }                                                                                                         \
    } while (0)
    if (row < rows) LN16_ISSUE(row);
    for (; row < rows; row += stride) {
        const float rs = rstd[row];
        const long ro = (long)row * D;
        float g[MAXC][4], xh[MAXC][4];
        bf16x4 rcur[MAXC];
        float s1 = 0.f, s2 = 0.f;
        asm volatile("" ::: "memory");       // the LDS parameter reads below are loop-invariant: keep them in the loop
#pragma unroll
        for (int i = 0; i < MAXC; ++i) rcur[i] = __builtin_nontemporal_load(reinterpret_cast<const bf16x4*>(dres16 + ro + off[i]));
        if (needx) {                         // (xh[] is not live yet: the fp32 rows land in its registers)
            const float mu = mean[row];
#pragma unroll
            for (int i = 0; i < MAXC; ++i)
                if (!((okmask >> i) & 1u)) {
                    load4_nt(x + ro + off[i], xh[i]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) xh[i][j] = (xh[i][j] - mu) * rs;
                }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + 64 * i;
            const bool valid = FULL || c < nch;
            const bool fromy = !needx || ((okmask >> i) & 1u);
            float gm[4], ig[4], bt[4];
            if (i) __builtin_amdgcn_sched_barrier(0);
            load4<float>(par + off[i], gm);
            load4<float>(par + D + off[i], ig);
            load4<float>(par + 2 * D + off[i], bt);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = valid ? (float)rdy[i][j] : 0.f;
                const float fy = ((float)ry[i][j] - bt[j]) * ig[j];
                xh[i][j] = valid ? (fromy ? fy : xh[i][j]) : 0.f;
                ag[i][j] += a * xh[i][j];
                ab[i][j] += a;
                g[i][j] = a * gm[j];
                s1 += g[i][j];
                s2 += g[i][j] * xh[i][j];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        { const int nr = row + stride < rows ? row + stride : row; LN16_ISSUE(nr); }      // (the last row re-reads itself: no branch, no copies)
        __builtin_amdgcn_sched_barrier(0);
        s1 = wave_sum_uniform(s1) / D;
        s2 = wave_sum_uniform(s2) / D;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + 64 * i;
            const bool valid = FULL || c < nch;
            float out[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) out[j] = rs * (g[i][j] - s1 - xh[i][j] * s2) + (float)rcur[i][j];
            if (drop_thresh) {
#pragma unroll
                for (int j = 0; j < 4; ++j) out[j] = dropout_keep(drop_seed, (unsigned long long)(ro + off[i]) + j, drop_thresh) ? out[j] * drop_scale : 0.f;
            }
            if (valid) {
                store4<bf16_t>(dx16 + ro + off[i], out);
#pragma unroll
                for (int j = 0; j < 4; ++j) as[i][j] += out[j];
            }
        }
    }
#undef LN16_ISSUE
    // cross-wave reduction of the column partials (as in ln_bwd_kernel)
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        __syncthreads();
        if (FULL || c < nch) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                red[0][wave * 256 + lane * 4 + j] = ag[i][j];
                red[1][wave * 256 + lane * 4 + j] = ab[i][j];
                red[2][wave * 256 + lane * 4 + j] = as[i][j];
            }
        }
        __syncthreads();
        if (wave == 0 && (FULL || c < nch)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float sg = 0.f, sb = 0.f, ss = 0.f;
                for (int w = 0; w < 4; ++w) { sg += red[0][w * 256 + lane * 4 + j]; sb += red[1][w * 256 + lane * 4 + j]; ss += red[2][w * 256 + lane * 4 + j]; }
                if (partials) {
                    float* pb = partials + (long)blockIdx.x * 3 * D + c * 4 + j;
                    pb[0] = sg; pb[D] = sb; pb[2 * D] = ss;
                } else {
                    atomicAdd(dgamma + c * 4 + j, sg);
                    atomicAdd(dbeta + c * 4 + j, sb);
                    if (dxsum) atomicAdd(dxsum + c * 4 + j, ss);
                }
            }
        }
    }
}

// Second stage of the LayerNorm backward column reductions.  With one atomic per column per block the 1024 blocks of a
// [100864, 768] launch sent 2.4 M atomics to 2304 addresses at the same moment (they finish together): 93 of the kernel's 307 us.
// Blocks now store their partial sums and this kernel folds them: thread = one of the 3*D columns, coalesced over columns.
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float* __restrict__ partials, int nblocks, int D, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, float* __restrict__ dxsum) {
    const int col = blockIdx.x * 256 + threadIdx.x;          // 0 .. 3*D
    if (col >= 3 * D) return;
    float* dst = col < D ? dgamma + col : (col < 2 * D ? dbeta + (col - D) : (dxsum ? dxsum + (col - 2 * D) : nullptr));
    if (!dst) return;
    const int per = (nblocks + gridDim.y - 1) / gridDim.y;   // partial rows of this y-slice: 32 independent loads in flight per thread
    const int b0 = blockIdx.y * per, b1 = min(nblocks, b0 + per);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = b0;
    for (; b + 4 <= b1; b += 4) {
        s0 += partials[(long)b * 3 * D + col];
        s1 += partials[(long)(b + 1) * 3 * D + col];
        s2 += partials[(long)(b + 2) * 3 * D + col];
        s3 += partials[(long)(b + 3) * 3 * D + col];
    }
    for (; b < b1; ++b) s0 += partials[(long)b * 3 * D + col];
    if (b1 > b0) atomicAdd(dst, (s0 + s1) + (s2 + s3));      // gridDim.y (32) adds per column instead of one per row-block (1024)
}

// ---------------------------------------------------------------------------------------------
// column sums  out[n] += sum_r in[r, n]   (bias gradients, token-type / position embedding gradients)
// block = 32 column-chunks (8 columns each) x 8 row lanes
// ---------------------------------------------------------------------------------------------
template <typename TI>
__global__ __launch_bounds__(256) void colsum_kernel(const TI* __restrict__ in, float* __restrict__ out, long rows, int N,
                                                     long ld, int rows_per_block) {
    __shared__ float red[8][32][8];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int col = (blockIdx.x * 32 + tx) * 8;
    const long r0 = (long)blockIdx.y * rows_per_block;
    const long r1 = min(rows, r0 + rows_per_block);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    if (col < N) {
        for (long r = r0 + ty; r < r1; r += 8) {
            float a[4], b[4];
            load4<TI>(in + r * ld + col, a);
            load4<TI>(in + r * ld + col + 4, b);
#pragma unroll
            for (int j = 0; j < 4; ++j) { acc[j] += a[j]; acc[4 + j] += b[j]; }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[ty][tx][j] = acc[j];
    __syncthreads();
    if (ty == 0 && col < N) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float s = 0.f;
            for (int w = 0; w < 8; ++w) s += red[w][tx][j];
            atomicAdd(out + col + j, s);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// ViT patch gather (timm PatchEmbed Conv2d k16 s16 as a GEMM): image [B,3,H,W] fp32 -> cols [B*N, 768],
// column order (c, kh, kw) = flattened Conv2d weight [D,3,16,16].  One thread per 4 consecutive kw.
// ---------------------------------------------------------------------------------------------
template <typename TO>
__global__ void im2col_kernel(const float* __restrict__ img, TO* __restrict__ cols, int B, int H, int W) {
    const int gw = W >> 4, gh = H >> 4;
    const long total = (long)B * gh * gw * 192;   // 768 / 4 chunks per patch
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % 192);
        const long pr = i / 192;                  // patch row index b * N + ph * gw + pw
        const int pw = (int)(pr % gw);
        const int ph = (int)((pr / gw) % gh);
        const int b = (int)(pr / ((long)gw * gh));
        const int c = ch >> 6, kh = (ch >> 2) & 15, kw = (ch & 3) * 4;
        float v[4];
        load4<float>(img + (((long)b * 3 + c) * H + ph * 16 + kh) * W + pw * 16 + kw, v);
        store4<TO>(cols + pr * 768 + ch * 4, v);
    }
}

// x[b, 0, :] = cls + pos[0]   (vit_builder.py:15-17; the patch rows are written by the patch GEMM epilogue)
__global__ void vit_cls_rows_kernel(const float* __restrict__ cls, const float* __restrict__ pos, float* __restrict__ x,
                                    int B, int T, int D) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * D) return;
    const int b = i / D, d = i % D;
    x[(long)b * T * D + d] = cls[d] + pos[d];
}
// backward: dcls += sum_b dx[b,0,:],  dpos[0] handled by the position column-sum over all rows
// (blockIdx.y: slices of 16 images - three blocks walking all 512 rows serially, each load a DRAM latency, took 195 us; one atomic per
//  column and slice)
__global__ void vit_cls_grad_kernel(const float* __restrict__ dx, float* __restrict__ dcls, int B, int T, int D) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    const int b0 = blockIdx.y * 16, b1 = min(B, b0 + 16);
    float s = 0.f;
#pragma unroll 8
    for (int b = b0; b < b1; ++b) s += dx[(long)b * T * D + d];
    atomicAdd(dcls + d, s);
}

// ---------------------------------------------------------------------------------------------
// BERT embeddings (HF BertEmbeddings): sum[b,l,:] = word[id] + pos[l] + type[0]     (LayerNorm runs next)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bert_embed_kernel(const long* __restrict__ ids, const float* __restrict__ word,
                                                         const float* __restrict__ pos, const float* __restrict__ type0,
                                                         float* __restrict__ out, int rows, int L, int D, int vocab) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    long id = ids[row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const int l = row % L;
    for (int c = lane; c < (D >> 2); c += 64) {
        float a[4], b[4], t[4], o[4];
        load4<float>(word + id * D + c * 4, a);
        load4<float>(pos + (long)l * D + c * 4, b);
        load4<float>(type0 + c * 4, t);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = a[j] + t[j] + b[j];
        store4<float>(out + (long)row * D + c * 4, o);
    }
}
// backward: dword[id] += dsum[row] for valid (unmasked) tokens.  Masked tokens carry an exactly-zero gradient
// (they are masked as keys and overwritten before pooling), so they are skipped.
__global__ __launch_bounds__(256) void bert_embed_bwd_kernel(const long* __restrict__ ids, const long* __restrict__ mask,
                                                             const float* __restrict__ dsum, float* __restrict__ dword,
                                                             int rows, int D, int vocab) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    if (mask && mask[row] == 0) return;
    long id = ids[row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    for (int c = lane; c < D; c += 64) atomicAdd(dword + id * D + c, dsum[(long)row * D + c]);
}

// ---------------------------------------------------------------------------------------------
// LoDA top-k pooling + L2norm  (components/pooling.py:52-65 + normalization.py:6-11)
// tok [B,N,P] -> emb [B,P] = l2norm(mean of the k largest over tokens, per channel).  Block per batch row,
// one thread per channel (coalesced across channels), top-k kept sorted in registers.
// ---------------------------------------------------------------------------------------------
constexpr int POOL_MAXK = 8;

__device__ __forceinline__ float block_sum(float v, float* sh) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < nw; ++i) s += sh[i];
    return s;
}

template <typename TI>
__global__ void topk_pool_fwd_kernel(const TI* __restrict__ tok, const long* __restrict__ mask, float* __restrict__ emb,
                                     int* __restrict__ idx, float* __restrict__ norm_out, int N, int P, int k, float eps, int normalize) {
    __shared__ float sh[16];
    const int b = blockIdx.x, c = threadIdx.x;
    float tv[POOL_MAXK];
    int ti[POOL_MAXK];
#pragma unroll
    for (int j = 0; j < POOL_MAXK; ++j) { tv[j] = -INFINITY; ti[j] = 0; }
    const TI* base = tok + (long)b * N * P + c;
    for (int n = 0; n < N; ++n) {
        float v = (float)base[(long)n * P];
        if (mask && mask[(long)b * N + n] == 0) v = -10000.f;        // pooling.py:60
        if (v > tv[k - 1]) {
            // insertion into the descending list (static indexing keeps tv/ti in registers)
            // once the insertion point is found everything below shifts down unconditionally, so among equal values the
            // LAST (largest index) falls off: the list is the top-k in (value descending, index ascending) order
            float cv = v; int ci = n;
            bool ins = false;
#pragma unroll
            for (int j = 0; j < POOL_MAXK; ++j) {
                if (j < k && (ins || cv > tv[j])) { const float t = tv[j]; const int u = ti[j]; tv[j] = cv; ti[j] = ci; cv = t; ci = u; ins = true; }
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < POOL_MAXK; ++j) if (j < k) { s += tv[j]; idx[((long)b * k + j) * P + c] = ti[j]; }
    const float pooled = s / k;
    if (!normalize) { emb[(long)b * P + c] = pooled; if (c == 0) norm_out[b] = 1.f; return; }
    const float nrm = sqrtf(block_sum(pooled * pooled, sh));
    emb[(long)b * P + c] = pooled / (nrm + eps);
    if (c == 0) norm_out[b] = nrm;
}

// Small-batch form of the same pooling.  One block per image leaves a single block scanning all N tokens when B = 1 (the
// reference tool's batch size): 554 us of a 2.5 ms ViT-B forward.  Here POOL_SLICES blocks per image each keep the top-k of a
// contiguous token slice (part A), and one block per image merges the POOL_SLICES*k candidates in (value desc, index asc)
// order - the order the single pass implies, so ties resolve identically - and normalises (part B).
constexpr int POOL_SLICES = 32;
template <typename TI>
__global__ void topk_pool_slice_kernel(const TI* __restrict__ tok, const long* __restrict__ mask, float* __restrict__ cv,
                                       int* __restrict__ ci, int N, int P, int k) {
    const int b = blockIdx.x, sl = blockIdx.y, c = threadIdx.x;
    const int chunk = (N + POOL_SLICES - 1) / POOL_SLICES;
    const int n0 = sl * chunk, n1 = min(N, n0 + chunk);
    float tv[POOL_MAXK];
    int ti[POOL_MAXK];
#pragma unroll
    for (int j = 0; j < POOL_MAXK; ++j) { tv[j] = -INFINITY; ti[j] = 0; }
    const TI* base = tok + (long)b * N * P + c;
    for (int nb = n0; nb < n1; nb += 8) {             // eight independent loads in flight, then the (serial) insertions
        float vs[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int n = nb + u;
            float v = -INFINITY;
            if (n < n1) {
                v = (float)base[(long)n * P];
                if (mask && mask[(long)b * N + n] == 0) v = -10000.f;
            }
            vs[u] = v;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (vs[u] > tv[k - 1]) {
                float x = vs[u]; int xi = nb + u;
                bool ins = false;
#pragma unroll
                for (int j = 0; j < POOL_MAXK; ++j)
                    if (j < k && (ins || x > tv[j])) { const float t = tv[j]; const int w = ti[j]; tv[j] = x; ti[j] = xi; x = t; xi = w; ins = true; }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < POOL_MAXK; ++j)
        if (j < k) {
            const long o = (((long)b * POOL_SLICES + sl) * k + j) * P + c;
            cv[o] = tv[j]; ci[o] = ti[j];
        }
}

// Merge of sorted candidate lists, in (value desc, index asc) order.  cv / ci: [B, L, k, P].  Block (b, g) merges lists
// g*LPG .. g*LPG + LPG - 1.  Two launches: POOL_SLICES lists -> POOL_GROUPS lists (FINAL = false: the merged list is written back
// out as [B, POOL_GROUPS, k, P]), then those -> the pooled, normalised embedding (FINAL = true).  One pass over all POOL_SLICES * k
// candidates in a single block per image was 133 us of a 1.5 ms batch-1 forward.
constexpr int POOL_GROUPS = 4;
template <bool FINAL>
__global__ void topk_pool_merge_kernel(const float* __restrict__ cv, const int* __restrict__ ci, float* __restrict__ ov, int* __restrict__ oi,
                                       float* __restrict__ emb, int* __restrict__ idx, float* __restrict__ norm_out, int L, int LPG,
                                       int P, int k, float eps, int normalize) {
    __shared__ float sh[16];
    const int b = blockIdx.x, g = blockIdx.y, c = threadIdx.x;
    float tv[POOL_MAXK];
    int ti[POOL_MAXK];
#pragma unroll
    for (int j = 0; j < POOL_MAXK; ++j) { tv[j] = -INFINITY; ti[j] = 0x7fffffff; }
    const int M = LPG * k;
    const long base = ((long)b * L + (long)g * LPG) * k;
    constexpr int MB = 8;                             // candidate loads in flight per round trip
    for (int mb = 0; mb < M; mb += MB) {
        float xs[MB];
        int is[MB];
#pragma unroll
        for (int u = 0; u < MB; ++u) {
            const long o = (base + min(mb + u, M - 1)) * P + c;
            xs[u] = mb + u < M ? cv[o] : -INFINITY;
            is[u] = ci[o];
        }
#pragma unroll
        for (int u = 0; u < MB; ++u) {
            float x = xs[u];
            int xi = is[u];
            if (x == -INFINITY) continue;             // empty slot of a short slice
            bool ins = false;
#pragma unroll
            for (int j = 0; j < POOL_MAXK; ++j)
                if (j < k && (ins || x > tv[j] || (x == tv[j] && xi < ti[j]))) { const float t = tv[j]; const int w = ti[j]; tv[j] = x; ti[j] = xi; x = t; xi = w; ins = true; }
        }
    }
    if (!FINAL) {
#pragma unroll
        for (int j = 0; j < POOL_MAXK; ++j)
            if (j < k) {
                const long o = (((long)b * gridDim.y + g) * k + j) * P + c;
                ov[o] = tv[j]; oi[o] = ti[j];
            }
        return;
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < POOL_MAXK; ++j) if (j < k) { s += tv[j]; idx[((long)b * k + j) * P + c] = ti[j]; }
    const float pooled = s / k;
    if (!normalize) { emb[(long)b * P + c] = pooled; if (c == 0) norm_out[b] = 1.f; return; }
    const float nrm = sqrtf(block_sum(pooled * pooled, sh));
    emb[(long)b * P + c] = pooled / (nrm + eps);
    if (c == 0) norm_out[b] = nrm;
}

// backward: demb [B,P] -> dtok [B,N,P] (dense, zero off the selected tokens)
template <typename TO>
__global__ void topk_pool_bwd_kernel(const float* __restrict__ demb, const float* __restrict__ emb,
                                     const float* __restrict__ norm, const int* __restrict__ idx, TO* __restrict__ dtok,
                                     int N, int P, int k, float eps, int normalize) {
    __shared__ float sh[16];
    const int b = blockIdx.x, c = threadIdx.x;
    const float g = demb[(long)b * P + c], y = emb[(long)b * P + c];
    const float n = norm[b];
    float dx = g / k;
    if (normalize) {
        const float dot = block_sum(g * y, sh);
        // y = x / (n + eps):  dx = [g - y (g.y)(n+eps)/n] / (n+eps)
        dx = (g - y * dot * (n + eps) / fmaxf(n, 1e-30f)) / (n + eps) / k;
    }
    int sel[POOL_MAXK];
#pragma unroll
    for (int j = 0; j < POOL_MAXK; ++j) sel[j] = j < k ? idx[((long)b * k + j) * P + c] : -1;
    TO* base = dtok + (long)b * N * P + c;
    for (int n_ = 0; n_ < N; ++n_) {
        bool hit = false;
#pragma unroll
        for (int j = 0; j < POOL_MAXK; ++j) hit |= (sel[j] == n_);
        base[(long)n_ * P] = (TO)(hit ? dx : 0.f);
    }
}

// Prompt-ensemble reduction of the zero-shot classifier (tools/seg_evaluation.py:71-73): out[s,:] = normalize(mean_p x[s,p,:])
__global__ void segment_mean_l2norm_kernel(const float* __restrict__ x, float* __restrict__ out, int Pn, int D) {
    __shared__ float sh[16];
    const int sgm = blockIdx.x, c = threadIdx.x;
    const float* base = x + (long)sgm * Pn * D + c;
    float acc = 0.f;
    for (int i = 0; i < Pn; ++i) acc += base[(long)i * D];
    acc /= Pn;
    const float nrm = sqrtf(block_sum(acc * acc, sh));
    out[(long)sgm * D + c] = acc / nrm;
}

// rnorm[row] = 1 / max(||x_row||, eps)   (F.normalize, tools/seg_evaluation.py:112); one wave per row
template <typename TI>
__global__ __launch_bounds__(256) void row_rnorm_kernel(const TI* __restrict__ x, float* __restrict__ rn, long rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float s = 0.f;
    for (int c = lane; c < (D >> 2); c += 64) {
        float v[4];
        load4<TI>(x + row * D + c * 4, v);
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    s = wave_sum(s);
    if (lane == 0) rn[row] = 1.0f / fmaxf(sqrtf(s), eps);
}

__global__ void cast_f32_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float v[4];
        load4<float>(in + i * 4, v);
        store4<bf16_t>(out + i * 4, v);
    }
}
__global__ void cast_bf16_f32_kernel(const bf16_t* __restrict__ in, float* __restrict__ out, long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float v[4];
        load4<bf16_t>(in + i * 4, v);
        store4<float>(out + i * 4, v);
    }
}

// out[c, r] = in[r, c]  (fp32, 32x32 LDS tiles)
__global__ void transpose_f32_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int C) {
    __shared__ float t[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int j = ty; j < 32; j += 8)
        if (r0 + j < R && c0 + tx < C) t[j][tx] = in[(long)(r0 + j) * C + c0 + tx];
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (c0 + j < C && r0 + tx < R) out[(long)(c0 + j) * R + r0 + tx] = t[tx][j];
}

// in-place dropout mask re-application for backward: g[i] = keep(i) ? g[i] * scale : 0
template <typename T>
__global__ void dropout_apply_kernel(T* __restrict__ g, long n, unsigned long long seed, unsigned int thresh, float scale) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        g[i] = (T)(dropout_keep(seed, (unsigned long long)i, thresh) ? (float)g[i] * scale : 0.f);
}

inline int grid_for(long n, int block, int cap = 4096) {
    long g = (n + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// One resident round of the LayerNorm backward: blocks per CU from the runtime's occupancy calculator for the instantiation's registers
// and LDS (cached per instantiation and LDS size), times the CU count.  Round 4 launched a fixed 1024 blocks; at 148 VGPRs 768 of them
// were resident and a quarter-full second round followed.
constexpr int LN_BWD_MAX_GRID = 2048;
inline int ln_bwd_grid(const void* fn, int slot, size_t lds, long rows) {
    // Host threads race here (autograd runs the two towers' backward passes on two threads), and a process may drive several devices: the
    // cache is per device, and an entry is ONE 64-bit word - (LDS bytes << 8 | blocks per CU), read and written atomically - so a reader
    // never pairs one call's occupancy with another call's LDS size.  A lost race costs a repeated query, never a wrong grid.
    constexpr int MAXDEV = 16;
    static std::atomic<int> ncu[MAXDEV];                       // CUs of the device, asked once (hipGetDeviceProperties is a millisecond-class call)
    static std::atomic<unsigned long long> cached[MAXDEV][32]; // per (device, instantiation slot)
    static std::atomic<int> forced{-1};                        // SIMSEG_LN_BWD_BLOCKS_PER_CU (tools/ln_bench.py sweeps), read once
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) dev = 0;
    int n_cu = ncu[dev].load(std::memory_order_relaxed);
    if (n_cu <= 0) {
        n_cu = 256;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) n_cu = prop.multiProcessorCount;
        ncu[dev].store(n_cu, std::memory_order_relaxed);
    }
    int f = forced.load(std::memory_order_relaxed);
    if (f < 0) {
        const char* e = getenv("SIMSEG_LN_BWD_BLOCKS_PER_CU");
        const int v = e ? atoi(e) : 0;
        f = v > 0 ? v : 0;
        forced.store(f, std::memory_order_relaxed);
    }
    const bool ok = slot >= 0 && slot < 32;
    const unsigned long long ent = ok ? cached[dev][slot].load(std::memory_order_relaxed) : 0ull;
    int per_cu = (ent != 0 && (ent >> 8) == (unsigned long long)lds) ? (int)(ent & 0xff) : 0;
    if (per_cu <= 0) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, lds) != hipSuccess || per_cu <= 0) { (void)hipGetLastError(); per_cu = 4; }
        if (per_cu > 255) per_cu = 255;
        if (ok) cached[dev][slot].store(((unsigned long long)lds << 8) | (unsigned long long)per_cu, std::memory_order_relaxed);
    }
    if (f > 0) per_cu = f;
    const long per_dev = (long)per_cu * n_cu;
    const int want = grid_for(rows, 4, LN_BWD_MAX_GRID);
    return want < per_dev ? want : (int)per_dev;
}


// ---- fp32 operand -> three bf16 pieces, laid out for ONE bf16 GEMM that reproduces the fp32 product -----------------------------
// x = hi + mid + lo with hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid): three round-to-nearest pieces of 8 significant
// bits each carry x's 24 (bf16 has fp32's exponent range, so the residuals are exact and in range).  A product a.b is then the sum
// of nine piece products; the three smallest (mid.lo, lo.mid, lo.lo <= 2^-24 |a||b|) are dropped, like the rounding of the fp32 FMA
// chain they replace.  The six kept ones are laid out along K - the A operand as [hi | hi | hi | mid | mid | lo], the B operand as
// [hi | mid | lo | hi | mid | hi], K' = 6 K - so that one launch of the bf16 MFMA kernel (v_mfma_f32_32x32x16_bf16, fp32 accumulate)
// forms all of them in its accumulators: six times the MFMA work at sixteen times the matrix rate of v_mfma_f32_32x32x2_f32.
// One thread per 8 consecutive k of a row: 32 B in, six 16-byte stores out.
__global__ __launch_bounds__(256) void split_bf16x3_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, long rows, int K, long ld_in, int bpat) {
    const int kc = K >> 3;
    const long total = rows * kc;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const long r = t / kc;
        const int k0 = (int)(t - r * kc) << 3;
        const float4 a = *reinterpret_cast<const float4*>(in + r * ld_in + k0), b = *reinterpret_cast<const float4*>(in + r * ld_in + k0 + 4);
        const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        union { bf16_t h[8]; u32x4 v; } hi, mid, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            hi.h[e] = (bf16_t)x[e];
            const float r1 = x[e] - (float)hi.h[e];
            mid.h[e] = (bf16_t)r1;
            lo.h[e] = (bf16_t)(r1 - (float)mid.h[e]);
        }
        if (bpat == 2) {                                 // three planes [piece][row][K] (the attention kernel's operand form)
            bf16_t* o = out + r * (long)K + k0;
            *reinterpret_cast<u32x4*>(o) = hi.v;
            *reinterpret_cast<u32x4*>(o + rows * (long)K) = mid.v;
            *reinterpret_cast<u32x4*>(o + 2 * rows * (long)K) = lo.v;
            continue;
        }
        bf16_t* o = out + r * (6L * K) + k0;
        // A pattern: hi hi hi mid mid lo;  B pattern: hi mid lo hi mid hi
        *reinterpret_cast<u32x4*>(o) = hi.v;
        *reinterpret_cast<u32x4*>(o + K) = bpat ? mid.v : hi.v;
        *reinterpret_cast<u32x4*>(o + 2L * K) = bpat ? lo.v : hi.v;
        *reinterpret_cast<u32x4*>(o + 3L * K) = bpat ? hi.v : mid.v;
        *reinterpret_cast<u32x4*>(o + 4L * K) = mid.v;
        *reinterpret_cast<u32x4*>(o + 5L * K) = bpat ? hi.v : lo.v;
    }
}

}  // namespace

#define STREAM ((hipStream_t)stream)

extern "C" int simseg_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y, int out_dtype,
                                    void* y_bf16, float* mean, float* rstd, int64_t rows, int64_t D, float eps, void* stream) {
    SS_HALF_FWD(simseg_layernorm_fwd, x, gamma, beta, y, out_dtype, y_bf16, mean, rstd, rows, D, eps, stream);
    SS_CHECK(x && gamma && beta && y, "layernorm_fwd: null pointer");
    SS_CHECK(D % 4 == 0 && D <= LN_MAXC * 256 && D > 0, "layernorm_fwd: D=%lld must be a multiple of 4 and <= %d", (long long)D, LN_MAXC * 256);
    if (rows <= 0) return 0;
    if (D <= 512) {                           // two rows per wave
        dim3 grid2((unsigned)((rows + 7) / 8));
        if (out_dtype == 0)
            hipLaunchKernelGGL(ln_fwd2_kernel<float>, grid2, dim3(256), 0, STREAM, x, gamma, beta, (float*)y, (bf16_t*)y_bf16, mean, rstd, (int)rows, (int)D, eps);
        else
            hipLaunchKernelGGL(ln_fwd2_kernel<bf16_t>, grid2, dim3(256), 0, STREAM, x, gamma, beta, (bf16_t*)y, (bf16_t*)y_bf16, mean, rstd, (int)rows, (int)D, eps);
        SS_LAUNCH_CHECK("layernorm_fwd");
        return 0;
    }
    dim3 grid((unsigned)((rows + 3) / 4));
    if (out_dtype == 0)
        hipLaunchKernelGGL(ln_fwd_kernel<float>, grid, dim3(256), 0, STREAM, x, gamma, beta, (float*)y, (bf16_t*)y_bf16, mean, rstd, (int)rows, (int)D, eps);
    else
        hipLaunchKernelGGL(ln_fwd_kernel<bf16_t>, grid, dim3(256), 0, STREAM, x, gamma, beta, (bf16_t*)y, (bf16_t*)y_bf16, mean, rstd, (int)rows, (int)D, eps);
    SS_LAUNCH_CHECK("layernorm_fwd");
    return 0;
}

extern "C" int simseg_layernorm_bwd(const void* dy_bf16, const float* dy_f32, const float* dres, const void* dres_bf16, const float* x,
                                    const void* y_bf16, const float* beta,
                                    const float* mean, const float* rstd, const float* gamma, float* dx_f32, void* dx_bf16,
                                    float* dgamma, float* dbeta, float* dxsum, float* partials, int64_t rows, int64_t D,
                                    uint64_t drop_seed, float drop_p, void* stream) {
    SS_HALF_FWD(simseg_layernorm_bwd, dy_bf16, dy_f32, dres, dres_bf16, x, y_bf16, beta, mean, rstd, gamma, dx_f32, dx_bf16, dgamma, dbeta, dxsum, partials, rows, D, drop_seed, drop_p, stream);
    SS_CHECK((dy_bf16 || dy_f32) && x && mean && rstd && gamma && dgamma && dbeta && (dx_f32 || dx_bf16) && (!y_bf16 || beta), "layernorm_bwd: null pointer");
    SS_CHECK(D % 4 == 0 && D <= LN_MAXC * 256 && D > 0, "layernorm_bwd: bad D=%lld", (long long)D);
    SS_CHECK(drop_p >= 0.f && drop_p < 1.f, "layernorm_bwd: dropout p out of range");
    if (rows <= 0) return 0;
    const unsigned int thresh = drop_p > 0.f ? (unsigned int)((double)drop_p * 4294967296.0) : 0u;
    const float scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    const int nc = (int)((D + 255) / 256);
    // the 16-bit ViT step's form - dy16 + y16 + dres16 -> dx16 and nothing in fp32 - has its own kernel
    const bool fast = dy_bf16 && y_bf16 && dres_bf16 && dx_bf16 && !dy_f32 && !dres && !dx_f32;
    const size_t lds = (size_t)(y_bf16 ? 3 : 1) * D * sizeof(float);
    int grid = 0;
#define LN_BWD_LAUNCH(C)                                                                                                              \
    do {                                                                                                                              \
        grid = ln_bwd_grid((const void*)ln_bwd_kernel<C>, C * 3, lds, rows);                                                          \
        hipLaunchKernelGGL(ln_bwd_kernel<C>, dim3(grid), dim3(256), lds, STREAM, (const bf16_t*)dy_bf16, dy_f32, dres, (const bf16_t*)dres_bf16, x, (const bf16_t*)y_bf16, beta, mean, rstd, gamma, \
                           dx_f32, (bf16_t*)dx_bf16, dgamma, dbeta, dxsum, partials, (int)rows, (int)D, (unsigned long long)drop_seed, thresh, scale); \
    } while (0)
#define LN_BWD16_LAUNCH(C, F)                                                                                                         \
    do {                                                                                                                              \
        grid = ln_bwd_grid((const void*)ln_bwd16_kernel<C, F>, C * 3 + (F ? 1 : 2), lds, rows);                                       \
        hipLaunchKernelGGL((ln_bwd16_kernel<C, F>), dim3(grid), dim3(256), lds, STREAM, (const bf16_t*)dy_bf16, (const bf16_t*)dres_bf16, x, (const bf16_t*)y_bf16, beta, mean, rstd, gamma, \
                           (bf16_t*)dx_bf16, dgamma, dbeta, dxsum, partials, (int)rows, (int)D, (unsigned long long)drop_seed, thresh, scale); \
    } while (0)
#define LN_BWD_PICK(C) do { if (!fast) LN_BWD_LAUNCH(C); else if (D == (C) * 256) LN_BWD16_LAUNCH(C, true); else LN_BWD16_LAUNCH(C, false); } while (0)
    if (nc <= 1) LN_BWD_PICK(1);
    else if (nc == 2) LN_BWD_PICK(2);
    else if (nc == 3) LN_BWD_PICK(3);
    else if (nc == 4) LN_BWD_PICK(4);
    else LN_BWD_PICK(8);
#undef LN_BWD_PICK
#undef LN_BWD16_LAUNCH
#undef LN_BWD_LAUNCH
    if (partials)
        hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((unsigned)((3 * D + 255) / 256), 32), dim3(256), 0, STREAM, partials, grid, (int)D,
                           dgamma, dbeta, dxsum);
    SS_LAUNCH_CHECK("layernorm_bwd");
    return 0;
}

/* floats the caller must provide as `partials` for a launch over `rows` rows of width D (the grid never exceeds LN_BWD_MAX_GRID blocks) */
extern "C" int64_t simseg_layernorm_bwd_workspace_bytes(int64_t rows, int64_t D) { return (int64_t)grid_for(rows, 4, LN_BWD_MAX_GRID) * 3 * D * (int64_t)sizeof(float); }

extern "C" int simseg_colsum_accum(const void* in, int in_dtype, float* out, int64_t rows, int64_t N, int64_t ld, void* stream) {
    SS_HALF_FWD(simseg_colsum_accum, in, in_dtype, out, rows, N, ld, stream);
    SS_CHECK(in && out, "colsum: null pointer");
    SS_CHECK(N % 8 == 0 && ld % 8 == 0, "colsum: N and ld must be multiples of 8");
    if (rows <= 0) return 0;
    int chunks = (int)((rows + 255) / 256);
    if (chunks > 512) chunks = 512;
    const int nbx = (int)((N + 255) / 256);
    if ((long)chunks * nbx < 512) {      // few rows (the [B, 3*H*64] partials of the attention backward: 18 blocks, each thread a chain of 32 loads - 100 us):
        const long by_rows = (rows + 7) / 8, by_grid = (512 + nbx - 1) / nbx;        // down to 8 rows per block until the grid fills the chip
        chunks = (int)(by_rows < by_grid ? by_rows : by_grid);
    }
    const int rpb = (int)((rows + chunks - 1) / chunks);
    dim3 grid((unsigned)nbx, (unsigned)((rows + rpb - 1) / rpb));
    if (in_dtype == 0)
        hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, STREAM, (const float*)in, out, (long)rows, (int)N, (long)ld, rpb);
    else
        hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, STREAM, (const bf16_t*)in, out, (long)rows, (int)N, (long)ld, rpb);
    SS_LAUNCH_CHECK("colsum");
    return 0;
}

extern "C" int simseg_vit_im2col(const float* image, void* cols, int out_dtype, int64_t B, int64_t H, int64_t W, void* stream) {
    SS_HALF_FWD(simseg_vit_im2col, image, cols, out_dtype, B, H, W, stream);
    SS_CHECK(image && cols, "im2col: null pointer");
    SS_CHECK(H % 16 == 0 && W % 16 == 0 && H > 0 && W > 0, "im2col: H, W must be multiples of the 16x16 patch");
    const long total = B * (H / 16) * (W / 16) * 192;
    if (total <= 0) return 0;
    const int grid = grid_for(total, 256, 8192);
    if (out_dtype == 0)
        hipLaunchKernelGGL(im2col_kernel<float>, dim3(grid), dim3(256), 0, STREAM, image, (float*)cols, (int)B, (int)H, (int)W);
    else
        hipLaunchKernelGGL(im2col_kernel<bf16_t>, dim3(grid), dim3(256), 0, STREAM, image, (bf16_t*)cols, (int)B, (int)H, (int)W);
    SS_LAUNCH_CHECK("im2col");
    return 0;
}

extern "C" int simseg_vit_cls_rows(const float* cls, const float* pos, float* x, int64_t B, int64_t T, int64_t D, void* stream) {
    SS_CHECK(cls && pos && x, "vit_cls_rows: null pointer");
    if (B <= 0) return 0;
    hipLaunchKernelGGL(vit_cls_rows_kernel, dim3((unsigned)((B * D + 255) / 256)), dim3(256), 0, STREAM, cls, pos, x, (int)B, (int)T, (int)D);
    SS_LAUNCH_CHECK("vit_cls_rows");
    return 0;
}

extern "C" int simseg_vit_cls_grad(const float* dx, float* dcls, int64_t B, int64_t T, int64_t D, void* stream) {
    SS_CHECK(dx && dcls, "vit_cls_grad: null pointer");
    hipLaunchKernelGGL(vit_cls_grad_kernel, dim3((unsigned)((D + 255) / 256), (unsigned)((B + 15) / 16)), dim3(256), 0, STREAM, dx, dcls, (int)B, (int)T, (int)D);
    SS_LAUNCH_CHECK("vit_cls_grad");
    return 0;
}

extern "C" int simseg_bert_embed_fwd(const int64_t* ids, const float* word, const float* pos, const float* type0, float* out,
                                     int64_t B, int64_t L, int64_t D, int64_t vocab, void* stream) {
    SS_CHECK(ids && word && pos && type0 && out, "bert_embed_fwd: null pointer");
    SS_CHECK(D % 4 == 0, "bert_embed_fwd: D must be a multiple of 4");
    const long rows = B * L;
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(bert_embed_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, STREAM, (const long*)ids, word, pos, type0, out,
                       (int)rows, (int)L, (int)D, (int)vocab);
    SS_LAUNCH_CHECK("bert_embed_fwd");
    return 0;
}

extern "C" int simseg_bert_embed_bwd(const int64_t* ids, const int64_t* mask, const float* dsum, float* dword, int64_t B,
                                     int64_t L, int64_t D, int64_t vocab, void* stream) {
    SS_CHECK(ids && dsum && dword, "bert_embed_bwd: null pointer");
    const long rows = B * L;
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(bert_embed_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, STREAM, (const long*)ids, (const long*)mask, dsum,
                       dword, (int)rows, (int)D, (int)vocab);
    SS_LAUNCH_CHECK("bert_embed_bwd");
    return 0;
}

extern "C" int64_t simseg_topk_pool_workspace_bytes(int64_t B, int64_t P, int k) { return B * (POOL_SLICES + POOL_GROUPS) * k * P * 2 * (int64_t)sizeof(float); }

extern "C" int simseg_topk_pool_l2norm_fwd(const void* tok, int dtype, const int64_t* mask, float* emb, int32_t* idx, float* norm,
                                           float* scratch, int64_t B, int64_t N, int64_t P, int k, float eps, int normalize,
                                           void* stream) {
    SS_HALF_FWD(simseg_topk_pool_l2norm_fwd, tok, dtype, mask, emb, idx, norm, scratch, B, N, P, k, eps, normalize, stream);
    SS_CHECK(tok && emb && idx && norm, "topk_pool_fwd: null pointer");
    SS_CHECK(P % 64 == 0 && P <= 1024, "topk_pool_fwd: P=%lld must be a multiple of 64 and <= 1024", (long long)P);
    SS_CHECK(k >= 1 && k <= POOL_MAXK && k <= N, "topk_pool_fwd: k=%d out of range (1..%d, <= N)", k, POOL_MAXK);
    if (B <= 0) return 0;
    if (scratch) {       // small batches: token slices in parallel, then a merge
        float* cv = scratch;
        int* ci = reinterpret_cast<int*>(scratch + B * POOL_SLICES * k * P);
        float* gv = scratch + 2 * B * POOL_SLICES * k * P;
        int* gi = reinterpret_cast<int*>(gv + B * POOL_GROUPS * k * P);
        dim3 g((unsigned)B, POOL_SLICES);
        if (dtype == 0)
            hipLaunchKernelGGL(topk_pool_slice_kernel<float>, g, dim3((unsigned)P), 0, STREAM, (const float*)tok, (const long*)mask, cv, ci, (int)N, (int)P, k);
        else
            hipLaunchKernelGGL(topk_pool_slice_kernel<bf16_t>, g, dim3((unsigned)P), 0, STREAM, (const bf16_t*)tok, (const long*)mask, cv, ci, (int)N, (int)P, k);
        hipLaunchKernelGGL(topk_pool_merge_kernel<false>, dim3((unsigned)B, POOL_GROUPS), dim3((unsigned)P), 0, STREAM, cv, ci, gv, gi, emb, idx, norm,
                           POOL_SLICES, POOL_SLICES / POOL_GROUPS, (int)P, k, eps, normalize);
        hipLaunchKernelGGL(topk_pool_merge_kernel<true>, dim3((unsigned)B, 1), dim3((unsigned)P), 0, STREAM, gv, gi, gv, gi, emb, idx, norm,
                           POOL_GROUPS, POOL_GROUPS, (int)P, k, eps, normalize);
        SS_LAUNCH_CHECK("topk_pool_fwd(sliced)");
        return 0;
    }
    if (dtype == 0)
        hipLaunchKernelGGL(topk_pool_fwd_kernel<float>, dim3((unsigned)B), dim3((unsigned)P), 0, STREAM, (const float*)tok, (const long*)mask, emb, idx, norm, (int)N, (int)P, k, eps, normalize);
    else
        hipLaunchKernelGGL(topk_pool_fwd_kernel<bf16_t>, dim3((unsigned)B), dim3((unsigned)P), 0, STREAM, (const bf16_t*)tok, (const long*)mask, emb, idx, norm, (int)N, (int)P, k, eps, normalize);
    SS_LAUNCH_CHECK("topk_pool_fwd");
    return 0;
}

extern "C" int simseg_topk_pool_l2norm_bwd(const float* demb, const float* emb, const float* norm, const int32_t* idx, void* dtok,
                                           int dtype, int64_t B, int64_t N, int64_t P, int k, float eps, int normalize, void* stream) {
    SS_HALF_FWD(simseg_topk_pool_l2norm_bwd, demb, emb, norm, idx, dtok, dtype, B, N, P, k, eps, normalize, stream);
    SS_CHECK(demb && emb && norm && idx && dtok, "topk_pool_bwd: null pointer");
    SS_CHECK(P % 64 == 0 && P <= 1024 && k >= 1 && k <= POOL_MAXK, "topk_pool_bwd: bad P/k");
    if (B <= 0) return 0;
    if (dtype == 0)
        hipLaunchKernelGGL(topk_pool_bwd_kernel<float>, dim3((unsigned)B), dim3((unsigned)P), 0, STREAM, demb, emb, norm, idx, (float*)dtok, (int)N, (int)P, k, eps, normalize);
    else
        hipLaunchKernelGGL(topk_pool_bwd_kernel<bf16_t>, dim3((unsigned)B), dim3((unsigned)P), 0, STREAM, demb, emb, norm, idx, (bf16_t*)dtok, (int)N, (int)P, k, eps, normalize);
    SS_LAUNCH_CHECK("topk_pool_bwd");
    return 0;
}

extern "C" int simseg_segment_mean_l2norm(const float* x, float* out, int64_t S, int64_t P, int64_t D, void* stream) {
    SS_CHECK(x && out, "segment_mean_l2norm: null pointer");
    SS_CHECK(D % 64 == 0 && D <= 1024 && P >= 1, "segment_mean_l2norm: D must be a multiple of 64 and <= 1024");
    if (S <= 0) return 0;
    hipLaunchKernelGGL(segment_mean_l2norm_kernel, dim3((unsigned)S), dim3((unsigned)D), 0, STREAM, x, out, (int)P, (int)D);
    SS_LAUNCH_CHECK("segment_mean_l2norm");
    return 0;
}

extern "C" int simseg_row_rnorm(const void* x, int dtype, float* rnorm, int64_t rows, int64_t D, float eps, void* stream) {
    SS_HALF_FWD(simseg_row_rnorm, x, dtype, rnorm, rows, D, eps, stream);
    SS_CHECK(x && rnorm, "row_rnorm: null pointer");
    SS_CHECK(D % 4 == 0, "row_rnorm: D must be a multiple of 4");
    if (rows <= 0) return 0;
    dim3 grid((unsigned)((rows + 3) / 4));
    if (dtype == 0)
        hipLaunchKernelGGL(row_rnorm_kernel<float>, grid, dim3(256), 0, STREAM, (const float*)x, rnorm, (long)rows, (int)D, eps);
    else
        hipLaunchKernelGGL(row_rnorm_kernel<bf16_t>, grid, dim3(256), 0, STREAM, (const bf16_t*)x, rnorm, (long)rows, (int)D, eps);
    SS_LAUNCH_CHECK("row_rnorm");
    return 0;
}

extern "C" int simseg_cast(const void* in, void* out, int64_t n, int to_bf16, void* stream) {
    SS_HALF_FWD(simseg_cast, in, out, n, to_bf16, stream);
    SS_CHECK(in && out, "cast: null pointer");
    SS_CHECK(n % 4 == 0, "cast: element count must be a multiple of 4");
    if (n <= 0) return 0;
    const int grid = grid_for(n / 4, 256, 8192);
    if (to_bf16)
        hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid), dim3(256), 0, STREAM, (const float*)in, (bf16_t*)out, (long)(n / 4));
    else
        hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(grid), dim3(256), 0, STREAM, (const bf16_t*)in, (float*)out, (long)(n / 4));
    SS_LAUNCH_CHECK("cast");
    return 0;
}

extern "C" int simseg_split_bf16x3(const float* in, void* out, int64_t rows, int64_t K, int64_t ld_in, int b_pattern, void* stream) {
    SS_CHECK(in && out, "split_bf16x3: null pointer");
    SS_CHECK(K > 0 && K % 8 == 0 && ld_in >= K && ld_in % 4 == 0, "split_bf16x3: K must be a multiple of 8 (got %lld)", (long long)K);
    SS_CHECK(((uintptr_t)in % 16) == 0 && ((uintptr_t)out % 16) == 0, "split_bf16x3: operands must be 16-byte aligned");
    if (rows <= 0) return 0;
    const int grid = grid_for(rows * (K / 8), 256, 16384);
    hipLaunchKernelGGL(split_bf16x3_kernel, dim3(grid), dim3(256), 0, STREAM, in, (bf16_t*)out, (long)rows, (int)K, (long)ld_in, b_pattern);
    SS_LAUNCH_CHECK("split_bf16x3");
    return 0;
}

extern "C" int simseg_transpose_f32(const float* in, float* out, int64_t R, int64_t C, void* stream) {
    SS_CHECK(in && out, "transpose: null pointer");
    if (R <= 0 || C <= 0) return 0;
    dim3 grid((unsigned)((C + 31) / 32), (unsigned)((R + 31) / 32));
    hipLaunchKernelGGL(transpose_f32_kernel, grid, dim3(256), 0, STREAM, in, out, (int)R, (int)C);
    SS_LAUNCH_CHECK("transpose");
    return 0;
}

// Row gather: dst[i,:] = src[idx[i],:] for i < n (idx < 0: a zero row).  Rows are `chunks` 16-byte pieces wide; dtype-agnostic.
// The text tower uses it to drop the padded token rows of a ragged caption batch before its GEMMs / LayerNorms and to put rows back
// (idx = inverse map, -1 at padded positions -> zeros) around the attention kernels, which keep the dense [B, L] layout.
static __global__ __launch_bounds__(256) void gather_rows_kernel(const u32x4* __restrict__ src, const int* __restrict__ idx, u32x4* __restrict__ dst,
                                                          long n, int chunks) {
    const long total = n * chunks;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const long i = t / chunks;
        const int c = (int)(t - i * chunks);
        const int r = idx[i];
        u32x4 v = {0u, 0u, 0u, 0u};
        if (r >= 0) v = src[(long)r * chunks + c];
        dst[t] = v;
    }
}

extern "C" int simseg_gather_rows(const void* src, const int32_t* idx, void* dst, int64_t n, int64_t row_bytes, void* stream) {
    SS_CHECK(src && idx && dst, "gather_rows: null pointer");
    SS_CHECK(row_bytes > 0 && row_bytes % 16 == 0 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0,
             "gather_rows: rows must be multiples of 16 bytes and 16-byte aligned");
    if (n <= 0) return 0;
    const long total = n * (row_bytes / 16);
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)grid_for(total, 256, 16384)), dim3(256), 0, STREAM, (const u32x4*)src, idx, (u32x4*)dst,
                       (long)n, (int)(row_bytes / 16));
    SS_LAUNCH_CHECK("gather_rows");
    return 0;
}

extern "C" int simseg_dropout_apply(void* g, int dtype, int64_t n, uint64_t seed, float p, void* stream) {
    SS_HALF_FWD(simseg_dropout_apply, g, dtype, n, seed, p, stream);
    SS_CHECK(g, "dropout_apply: null pointer");
    SS_CHECK(p >= 0.f && p < 1.f, "dropout_apply: p out of range");
    if (n <= 0 || p == 0.f) return 0;
    const unsigned int thresh = (unsigned int)((double)p * 4294967296.0);
    const float scale = 1.0f / (1.0f - p);
    const int grid = grid_for(n, 256, 8192);
    if (dtype == 0)
        hipLaunchKernelGGL(dropout_apply_kernel<float>, dim3(grid), dim3(256), 0, STREAM, (float*)g, (long)n, (unsigned long long)seed, thresh, scale);
    else
        hipLaunchKernelGGL(dropout_apply_kernel<bf16_t>, dim3(grid), dim3(256), 0, STREAM, (bf16_t*)g, (long)n, (unsigned long long)seed, thresh, scale);
    SS_LAUNCH_CHECK("dropout_apply");
    return 0;
}
