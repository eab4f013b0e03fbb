// Fused softmax attention (K5 of SURVEY.md 2.2) for head_dim 64: timm Attention (no mask, scale 64^-0.5) and
// HF BertSelfAttention (key-padding mask, optional attention-prob dropout).  Flash-style: the T x T score matrix is
// never written to HBM; K/V tiles of 64 keys are staged in LDS, softmax runs online in registers.
//
// Input is the packed projection qkv[B,T,3,H,64] (timm's qkv Linear layout; BERT's query/key/value weights are
// stored adjacently so the same single GEMM produces it).  Output ctx[B,T,H*64] is what the out-projection consumes.
//
// MFMA orientation ("swapped QK^T"): S^T = K.Q^T is computed with keys as MFMA rows and queries as MFMA columns, so a
// lane owns ONE query column (lane%32) and 16 of every 32 keys: the row max / row sum are in-lane reductions plus a
// single exchange with lane^32.  The PV product is taken as O^T = V^T.P^T so that the accumulator is again indexed
// by query per lane (the online-softmax rescale is lane-local) and P feeds the MFMA B operand straight from the
// score registers; V^T fragments come from the row-major V tile through ds_read_b64_tr_b16 (bf16) or plain b32
// reads (fp32).  fp32 path = v_mfma_f32_32x32x2_f32 (exact), bf16 path = v_mfma_f32_32x32x16_bf16.
#include "common.h"
#include <type_traits>

namespace {

constexpr int KT = 64;            // keys per LDS tile
constexpr float NEG = -1e30f;

struct AttnParams {
    const void* qkv; const long* mask; void* out; float* lse;
    int B, T, H;
    float scale_log2e;
    unsigned long long drop_seed; unsigned int drop_thresh; float drop_scale;
    // backward
    const void* dout; const float* delta; void* dqkv;
    int dbg;            // resident kernels, benchmarking only: 2 = skip the K/V copy, 3 = skip the tile loop
    unsigned long long* trace;   // debugging (simseg_debug_attn_trace): per block {start, operands landed, end} wall-clock stamps
    int pack;           // resident kernels: rows past the last unmasked key of a sequence are neither read nor written (see attn_teff)
    float* colsum_ws;   // one-kernel backward: [B][3*H*64] per-sequence column sums of dqkv (the qkv bias gradient before the fold over B), or null
    const int* row_start;   // resident forward / one-kernel backward: ragged batch stored WITHOUT its padding - sequence b occupies rows
                            // [row_start[b], row_start[b+1]) of qkv / out / dout / dqkv, every stored token is real; null = dense [B, T] rows
    long rs, ps;        // resident forward / one-kernel backward: element strides of qkv / dqkv - row r of plane (which * H + h), which = 0 q, 1 k, 2 v,
                        // starts at plane * ps + r * rs.  Packed projection rows [rows, 3, H, 64]: rs = 3 * H * 64, ps = 64; plane-major
                        // [3 * H][rows][64] (a head's operand rows are ONE contiguous run): rs = 64, ps = rows * 64
    int qscaled;        // w64 forward: the q columns of qkv already carry scale * log2(e) (simseg_attention_fwd_qscaled)
    int gxm, gxw;       // w64 forward: main blocks / all blocks per (batch, head)
    int pf_stride;      // one-kernel backward: > 0 = touch the operands of head blockIdx.x + pf_stride (the head that takes this CU's place in the
                        // next round) during the tile loop, so that its copies find them in this XCD's L2; 0 = off
};

// online softmax update for one 64-key tile; s[kb][r] holds raw scores for key kb*32 + (r%4) + 8*(r/4) + 4*(lane/32)
// FAST: raw v_exp_f32 (arguments are <= 0, results in [0,1]; no denormal-range fix-up code) -- used by the bf16 kernels,
// where the softmax VALU work, not the MFMAs, bounds the kernel
template <bool FAST> __device__ __forceinline__ float ex2(float x) {
    if constexpr (FAST) return __builtin_amdgcn_exp2f(x);
    else return exp2f(x);
}

// Block -> (head, row-block) map of every attention kernel.  The launch is 1-D; hardware hands consecutive block ids to
// consecutive XCDs, so id L runs on XCD L % 8.  All row-blocks of one (batch, head) are given ids of the same residue: the
// head's K/V (or Q/dO) is then pulled into ONE XCD's L2 and shared by its 2-6 blocks instead of being fetched by each
// block's own XCD (measured: 3-4 TB/s of L2-miss traffic and waves stalled on tile arrival before this map).
__device__ __forceinline__ bool attn_block_map(const AttnParams& p, int& qb, int& bh) {
    const int nw = blockDim.x >> 6;
    const int gx = ((p.T + 31) / 32 + nw - 1) / nw;
    const int L = blockIdx.x, slot = L >> 3;
    bh = (slot / gx) * 8 + (L & 7);
    qb = slot % gx;
    return bh < p.B * p.H;
}

// BIAS: additive per-key bias read from LDS (a key-padding mask was given).  Otherwise only the tile's padding (keys >= klim)
// is excluded, by comparison: no LDS reads, no registers for the bias.
template <bool FAST, bool BIAS = true>
__device__ __forceinline__ void softmax_tile(f32x16 (&s)[2], const float* kbias, int h2, float scale_log2e, float& m,
                                             float& lsum, float& alpha, int klim = 64) {
    float mx = NEG;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2;
            float x;
            if (BIAS) x = s[kb][r] * scale_log2e + kbias[key];
            else x = key < klim ? s[kb][r] * scale_log2e : NEG;
            s[kb][r] = x;
            mx = fmaxf(mx, x);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mn = fmaxf(m, mx);
    alpha = ex2<FAST>(m - mn);
    m = mn;
    float ps = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = ex2<FAST>(s[kb][r] - mn);
            s[kb][r] = p;
            ps += p;
        }
    lsum = lsum * alpha + ps;
}

// Lean online-softmax update for the bf16 kernels, where the softmax VALU work - not the MFMAs - sets the pace (a 64-key tile
// is 16 MFMAs = 512 matrix-pipe cycles per wave against ~300 VALU instructions in softmax_tile).  Three savings:
//   * interior tiles (no key bias, no padding) take the maximum of the RAW scores and fold the scale into the exponent's FMA:
//     one max, one FMA, one exp2 and one add per score - no compare / select, no separate multiply and subtract;
//   * the running maximum is only raised (and O, l rescaled) when some row's tile maximum exceeds it by more than RESCALE_THR
//     (log2 units): probabilities then reach at most 2^THR, which bf16 (constant relative precision) and the fp32 accumulators
//     carry without loss of accuracy, and the 32-multiply rescale of O disappears from almost every tile.  The decision is
//     wave-uniform, every quantity at the old maximum (O, l) is rescaled exactly once, P is formed after the decision;
//   * maxima and sums run as four independent chains.
// `o` holds the UNNORMALISED output accumulators; on return s holds P (to be multiplied into o by the caller).
constexpr float RESCALE_THR = 6.0f;

// `two` (wave-uniform; EDGE tiles only): false = the tile's second 32-key block is all padding - its scores were not formed, its
// probabilities are exactly 0 and nobody reads them, so its share of the arithmetic (half of an edge tile's ~150 VALU instructions: at
// T = 197 the last tile holds 5 keys, at T = 1025 one) is skipped.
template <bool BIAS, bool EDGE>
__device__ __forceinline__ void softmax_tile_lean(f32x16 (&s)[2], const float* kbias, int h2, float c, float& m, float& lsum,
                                                  f32x16 (&o)[2], int klim, bool two = true) {
    float mx[4] = {NEG, NEG, NEG, NEG};
    constexpr bool scaled = BIAS || EDGE;          // s is rewritten as the scaled, biased / masked score
    if (scaled) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            if (EDGE && kb == 1 && !two) break;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2;
                float x;
                if (BIAS) x = fmaf(s[kb][r], c, kbias[key]);
                else x = key < klim ? s[kb][r] * c : NEG;
                s[kb][r] = x;
                mx[r & 3] = fmaxf(mx[r & 3], x);
            }
        }
    } else {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx[r & 3] = fmaxf(mx[r & 3], s[kb][r]);
    }
    float tmax = fmaxf(fmaxf(mx[0], mx[1]), fmaxf(mx[2], mx[3]));
    if (!scaled) tmax *= c;                         // c > 0
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    if (!__all(tmax <= m + RESCALE_THR)) {          // wave-uniform
        const float mn = fmaxf(m, tmax);
        const float alpha = __builtin_amdgcn_exp2f(m - mn);
        m = mn;
        lsum *= alpha;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
    }
    float ps[4] = {0.f, 0.f, 0.f, 0.f};
    const float nm = -m;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        if (EDGE && kb == 1 && !two) break;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pr = __builtin_amdgcn_exp2f(scaled ? s[kb][r] + nm : fmaf(s[kb][r], c, nm));
            s[kb][r] = pr;
            ps[r & 3] += pr;
        }
    }
    lsum += (ps[0] + ps[1]) + (ps[2] + ps[3]);
}

// ------------------------------------------------------------------------------------------------
// fp32 forward
// ------------------------------------------------------------------------------------------------
constexpr int KP32 = 272, VP32 = 256;

__global__ __launch_bounds__(512) void attn_fwd_f32_kernel(AttnParams p) {
    __shared__ __attribute__((aligned(16))) char lds[KT * KP32 + KT * VP32 + KT * 4];
    char* ldsK = lds;
    char* ldsV = lds + KT * KP32;
    float* kbias = reinterpret_cast<float*>(lds + KT * KP32 + KT * VP32);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h2 = lane >> 5, ql = lane & 31;
    int qb_, bh_;
    if (!attn_block_map(p, qb_, bh_)) return;
    const int b = bh_ / p.H, h = bh_ % p.H;
    const int T = p.T;
    const long RS = 3L * p.H * 64;
    const float* base = static_cast<const float*>(p.qkv) + (long)b * T * RS + h * 64;
    const int nthr = blockDim.x;
    const int q = qb_ * (nthr >> 1) + wave * 32 + ql;

    float qr[8][4];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < T) v = *reinterpret_cast<const float4*>(base + (long)q * RS + 8 * c + 4 * h2);
        qr[c][0] = v.x; qr[c][1] = v.y; qr[c][2] = v.z; qr[c][3] = v.w;
    }
    f32x16 o[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float m = NEG, lsum = 0.f;

    for (int kv0 = 0; kv0 < T; kv0 += KT) {
        __syncthreads();
        for (int idx = tid; idx < KT * 16; idx += nthr) {      // 64 keys x 16 chunks of 16 B, for K and V
            const int key = idx >> 4, c = idx & 15;
            float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
            if (kv0 + key < T) {
                const float* rp = base + (long)(kv0 + key) * RS + c * 4;
                kv = *reinterpret_cast<const float4*>(rp + p.H * 64);
                vv = *reinterpret_cast<const float4*>(rp + 2 * p.H * 64);
            }
            *reinterpret_cast<float4*>(ldsK + key * KP32 + c * 16) = kv;
            *reinterpret_cast<float4*>(ldsV + key * VP32 + c * 16) = vv;
        }
        if (tid < KT) {
            const int key = kv0 + tid;
            kbias[tid] = (key < T && (!p.mask || p.mask[(long)b * T + key] != 0)) ? 0.f : NEG;
        }
        __syncthreads();

        f32x16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
            if (kv0 + kb * 32 >= T) continue;       // whole 32-key block is padding (kbias = NEG makes its P exactly 0)
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float4 a = *reinterpret_cast<const float4*>(ldsK + (kb * 32 + ql) * KP32 + (8 * c + 4 * h2) * 4);
                s[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, qr[c][0], s[kb], 0, 0, 0);
                s[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, qr[c][1], s[kb], 0, 0, 0);
                s[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, qr[c][2], s[kb], 0, 0, 0);
                s[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, qr[c][3], s[kb], 0, 0, 0);
            }
        }
        float alpha;
        softmax_tile<false>(s, kbias, h2, p.scale_log2e, m, lsum, alpha);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
        if (p.drop_thresh) {      // HF attention_probs dropout (same element index and hash as the bf16 kernels)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kv0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2;
                    const unsigned idx = ((unsigned)bh_ * T + q) * T + key;
                    s[kb][r] = dropout_keep32(seed_fold(p.drop_seed), idx, p.drop_thresh) ? s[kb][r] * p.drop_scale : 0.f;
                }
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            if (kv0 + kb * 32 >= T) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2;
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const float v = *reinterpret_cast<const float*>(ldsV + key * VP32 + (db * 32 + ql) * 4);
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(v, s[kb][r], o[db], 0, 0, 0);
                }
            }
        }
    }
    const float ltot = lsum + __shfl_xor(lsum, 32, 64);
    const float inv = 1.0f / ltot;
    if (q < T) {
        float* orow = static_cast<float*>(p.out) + ((long)b * T + q) * p.H * 64 + h * 64;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int d = db * 32 + 8 * r4 + 4 * h2;
                *reinterpret_cast<float4*>(orow + d) = make_float4(o[db][4 * r4] * inv, o[db][4 * r4 + 1] * inv, o[db][4 * r4 + 2] * inv, o[db][4 * r4 + 3] * inv);
            }
        if (p.lse && h2 == 0) p.lse[((long)b * p.H + h) * T + q] = m + log2f(ltot);
    }
}

// ------------------------------------------------------------------------------------------------
// fp32 backward (exact mode).  The reference trains in fp32 whenever cfg.dist.fp16 is off (simseg/core/config.py:50,
// simseg/core/hooks/optimizer.py:76-77); this is also the arithmetic the hand-written bf16 backward is checked against.
// One kernel body, two roles, same MFMA layout as attn_fwd_f32_kernel (lane column = an "own" row kept in registers, the
// other side streamed through LDS 64 rows at a time):
//   DKV = false: own = 32 queries (Q, dO in registers), streamed = keys (K, V):   dQ = scale * dS . K     (+ writes delta)
//   DKV = true : own = 32 keys (K, V in registers), streamed = queries (Q, dO):   dK = scale * dS^T . Q,  dV = Pd^T . dO
// with P = exp2(s * scale * log2e + keybias - lse), Pd = dropout(P), dS = P o (dropout(dO . V^T) - delta), delta = rowsum(dO o O).
// ------------------------------------------------------------------------------------------------
template <bool DKV, bool DROP>
__global__ __launch_bounds__(256) void attn_bwd_f32_kernel(AttnParams p) {
    __shared__ __attribute__((aligned(16))) char lds[2 * KT * KP32 + 2 * KT * 4];
    char* lds1 = lds;                    // K (dQ role) / Q (dK,dV role)
    char* lds2 = lds + KT * KP32;        // V            / dO
    float* rowa = reinterpret_cast<float*>(lds + 2 * KT * KP32);     // key bias / lse of the streamed queries
    float* rowb = rowa + KT;                                         //          / delta of the streamed queries
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h2 = lane >> 5, ql = lane & 31;
    int qb_, bh_;
    if (!attn_block_map(p, qb_, bh_)) return;
    const int b = bh_ / p.H, h = bh_ % p.H;
    const int T = p.T;
    const long HD = (long)p.H * 64, RS = 3 * HD;
    const float* base = static_cast<const float*>(p.qkv) + (long)b * T * RS + h * 64;
    const float* dob = static_cast<const float*>(p.dout) + (long)b * T * HD + h * 64;
    const float* ob = static_cast<const float*>(p.out) + (long)b * T * HD + h * 64;
    const long stat = ((long)b * p.H + h) * T;
    const int nthr = blockDim.x;
    const int own = qb_ * (nthr >> 1) + wave * 32 + ql;
    const float c_ = p.scale_log2e, scale = p.scale_log2e * 0.6931471805599453f;

    float a[8][4], g[8][4];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float4 u = make_float4(0.f, 0.f, 0.f, 0.f), v = u;
        if (own < T) {
            const int d = 8 * c + 4 * h2;
            u = *reinterpret_cast<const float4*>(base + (long)own * RS + (DKV ? HD : 0) + d);
            v = DKV ? *reinterpret_cast<const float4*>(base + (long)own * RS + 2 * HD + d)
                    : *reinterpret_cast<const float4*>(dob + (long)own * HD + d);
        }
        a[c][0] = u.x; a[c][1] = u.y; a[c][2] = u.z; a[c][3] = u.w;
        g[c][0] = v.x; g[c][1] = v.y; g[c][2] = v.z; g[c][3] = v.w;
    }
    float lse_own = 1e30f, delta_own = 0.f, bias_own = NEG;
    if (!DKV) {
        float dl = 0.f;
        if (own < T) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float4 o4 = *reinterpret_cast<const float4*>(ob + (long)own * HD + 8 * c + 4 * h2);
                dl += o4.x * g[c][0] + o4.y * g[c][1] + o4.z * g[c][2] + o4.w * g[c][3];
            }
        }
        dl += __shfl_xor(dl, 32, 64);
        delta_own = dl;
        if (own < T) {
            lse_own = p.lse[stat + own];
            if (h2 == 0) const_cast<float*>(p.delta)[stat + own] = dl;
        }
    } else if (own < T && (!p.mask || p.mask[(long)b * T + own] != 0)) bias_own = 0.f;

    f32x16 acc1[2], acc2[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc1[i][r] = 0.f; acc2[i][r] = 0.f; }

    for (int j0 = 0; j0 < T; j0 += KT) {
        __syncthreads();
        for (int idx = tid; idx < KT * 16; idx += nthr) {
            const int row = idx >> 4, c = idx & 15;
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f), y = x;
            if (j0 + row < T) {
                const float* rp = base + (long)(j0 + row) * RS + c * 4;
                if (DKV) { x = *reinterpret_cast<const float4*>(rp); y = *reinterpret_cast<const float4*>(dob + (long)(j0 + row) * HD + c * 4); }
                else { x = *reinterpret_cast<const float4*>(rp + HD); y = *reinterpret_cast<const float4*>(rp + 2 * HD); }
            }
            *reinterpret_cast<float4*>(lds1 + row * KP32 + c * 16) = x;
            *reinterpret_cast<float4*>(lds2 + row * KP32 + c * 16) = y;
        }
        if (tid < KT) {
            const int r = j0 + tid;
            if (DKV) { rowa[tid] = r < T ? p.lse[stat + r] : 1e30f; rowb[tid] = r < T ? p.delta[stat + r] : 0.f; }
            else rowa[tid] = (r < T && (!p.mask || p.mask[(long)b * T + r] != 0)) ? 0.f : NEG;
        }
        __syncthreads();
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            if (j0 + jb * 32 >= T) continue;
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float4 x = *reinterpret_cast<const float4*>(lds1 + (jb * 32 + ql) * KP32 + (8 * c + 4 * h2) * 4);
                const float4 y = *reinterpret_cast<const float4*>(lds2 + (jb * 32 + ql) * KP32 + (8 * c + 4 * h2) * 4);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(x.x, a[c][0], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x2f32(y.x, g[c][0], dp, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(x.y, a[c][1], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x2f32(y.y, g[c][1], dp, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(x.z, a[c][2], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x2f32(y.z, g[c][2], dp, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(x.w, a[c][3], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x2f32(y.w, g[c][3], dp, 0, 0, 0);
            }
            // s <- dS * scale, dp <- dropout(P)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = jb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2;
                float pr, dl;
                if (DKV) { pr = exp2f(fmaf(s[r], c_, bias_own) - rowa[row]); dl = rowb[row]; }
                else { pr = exp2f(fmaf(s[r], c_, rowa[row]) - lse_own); dl = delta_own; }
                float keep = 1.f;
                if (DROP) {
                    const unsigned qi = DKV ? j0 + row : own, ki = DKV ? own : j0 + row;
                    const unsigned idx = ((unsigned)bh_ * T + qi) * T + ki;                 // < 2^32: checked by the host
                    keep = dropout_keep32(seed_fold(p.drop_seed), idx, p.drop_thresh) ? p.drop_scale : 0.f;
                }
                s[r] = pr * (dp[r] * keep - dl) * scale;
                dp[r] = pr * keep;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = jb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2;
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const float x = *reinterpret_cast<const float*>(lds1 + row * KP32 + (db * 32 + ql) * 4);
                    acc1[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, s[r], acc1[db], 0, 0, 0);
                    if (DKV) {
                        const float y = *reinterpret_cast<const float*>(lds2 + row * KP32 + (db * 32 + ql) * 4);
                        acc2[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(y, dp[r], acc2[db], 0, 0, 0);
                    }
                }
            }
        }
    }
    if (own < T) {
        float* dst = static_cast<float*>(p.dqkv) + ((long)b * T + own) * RS + h * 64 + (DKV ? HD : 0);
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int d = db * 32 + 8 * r4 + 4 * h2;
                *reinterpret_cast<float4*>(dst + d) = make_float4(acc1[db][4 * r4], acc1[db][4 * r4 + 1], acc1[db][4 * r4 + 2], acc1[db][4 * r4 + 3]);
                if (DKV)
                    *reinterpret_cast<float4*>(dst + HD + d) = make_float4(acc2[db][4 * r4], acc2[db][4 * r4 + 1], acc2[db][4 * r4 + 2], acc2[db][4 * r4 + 3]);
            }
    }
}

// ------------------------------------------------------------------------------------------------
// bf16 forward
// ------------------------------------------------------------------------------------------------
constexpr int KP16 = 144, VP16 = 192;
typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;

__device__ __forceinline__ bf16x8 tr_frag(const char* p0, int row_step_bytes) {
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p0 + row_step_bytes));
    union { struct { s16x4 lo, hi; } s; bf16x8 v; } u;
    u.s.lo = lo; u.s.hi = hi;
    return u.v;
}

__device__ __forceinline__ bf16x8 ld_bf16x8(const void* p) {
    union { u32x4 v; bf16x8 h; } u;
    u.v = *reinterpret_cast<const u32x4*>(p);
    return u.h;
}

// K/V tile staging for the bf16 kernels: tile 0 goes global -> LDS directly; every later tile is fetched into registers
// while the previous tile is being consumed (later tiles exist only when T > 64, i.e. the block has >= 3 waves, so three
// 16-byte pieces per thread always cover the 512 pieces of a K or V tile).
struct KVRegs { u32x4 k[3], v[3]; };

__device__ __forceinline__ void kv_fetch(KVRegs& r, const bf16_t* base, long RS, int HD, int kv0, int T, int tid, int nthr) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int idx = tid + i * nthr;
        u32x4 kv = {0u, 0u, 0u, 0u}, vv = kv;
        if (idx < KT * 8) {
            const int key = idx >> 3, c = idx & 7;
            if (kv0 + key < T) {
                const bf16_t* rp = base + (long)(kv0 + key) * RS + c * 8;
                kv = *reinterpret_cast<const u32x4*>(rp + HD);
                vv = *reinterpret_cast<const u32x4*>(rp + 2 * HD);
            }
        }
        r.k[i] = kv; r.v[i] = vv;
    }
}

__device__ __forceinline__ void kv_commit(const KVRegs& r, char* ldsK, char* ldsV, int vpitch, int tid, int nthr) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int idx = tid + i * nthr;
        if (idx < KT * 8) {
            const int key = idx >> 3, c = idx & 7;
            *reinterpret_cast<u32x4*>(ldsK + key * KP16 + c * 16) = r.k[i];
            *reinterpret_cast<u32x4*>(ldsV + key * vpitch + c * 16) = r.v[i];
        }
    }
}

__device__ __forceinline__ void kv_direct(const bf16_t* base, long RS, int HD, int kv0, int T, char* ldsK, char* ldsV, int vpitch,
                                          int tid, int nthr) {
    for (int idx = tid; idx < KT * 8; idx += nthr) {
        const int key = idx >> 3, c = idx & 7;
        u32x4 kv = {0u, 0u, 0u, 0u}, vv = kv;
        if (kv0 + key < T) {
            const bf16_t* rp = base + (long)(kv0 + key) * RS + c * 8;
            kv = *reinterpret_cast<const u32x4*>(rp + HD);
            vv = *reinterpret_cast<const u32x4*>(rp + 2 * HD);
        }
        *reinterpret_cast<u32x4*>(ldsK + key * KP16 + c * 16) = kv;
        *reinterpret_cast<u32x4*>(ldsV + key * vpitch + c * 16) = vv;
    }
}

__device__ __forceinline__ void kbias_fill(float* kbias, const long* mask, int b, int kv0, int T, int tid) {
    if (tid < KT) {
        const int key = kv0 + tid;
        kbias[tid] = (key < T && (!mask || mask[(long)b * T + key] != 0)) ? 0.f : NEG;
    }
}

// DROP: attention_probs dropout compiled in (BERT training only).  MASK: a key-padding mask is given (BERT).  DBG: cycle-counter
// timeline written through p.delta (tools/dbg_attn_timeline.py); the production instantiations do not carry it.
//
// K/V tiles of 64 keys go global -> LDS directly (global_load_lds_dwordx4: no VGPR staging, no ds_write) into a ring of three
// stages, two tiles ahead of the one being consumed - the timeline showed ~5.5 k cycles of global latency under load against
// ~2.7 k cycles of work per tile, so one tile of lookahead was not enough.  One barrier per tile.  The LDS image of a
// direct load is lane-linear (1 KiB = 8 key rows of 128 B per wave-instruction), so the rows are unpadded and the bank
// swizzle is applied to the SOURCE chunk: K slot c of row r holds chunk c ^ ((r>>1)&7) (conflict-free ds_read_b128),
// V slot c holds chunk c ^ (((r>>1)&1)<<2) (conflict-free ds_read_b64_tr_b16).  Keys past T re-read row T-1 (finite
// values; their probabilities are exactly 0).
constexpr int GSTAGE = 2 * KT * 128;     // K then V
constexpr int GNS = 3;
constexpr int MAXT_BIAS = 1088;          // key-bias table of a masked sequence (BERT: T <= 512)

__device__ __forceinline__ int k_swz(int r) { return (r >> 1) & 7; }
__device__ __forceinline__ int v_swz(int r) { return ((r >> 1) & 1) << 2; }

// One global -> LDS copy (16 B per lane, 1 KiB per wave) of the ring kernels.  Through the builtin the compiler knows that LDS writes are
// in flight and puts `s_waitcnt vmcnt(0)` in front of the next LDS read it cannot prove disjoint - in these kernels the transposed V reads
// of the SAME iteration that issued the copies of a tile two ahead (visible in the ISA).  The inline-asm form (-DSS_GLDS_ASM) is invisible
// to that wait counting, the kernels' own vmcnt waits at the tile barriers being the only ones - and measured NO better: exact-mode kernel
// unchanged (1.33 ms), bf16 streaming forward at T = 1025 2.6 % slower (0.1010 vs 0.0983 ms, same-box A/B of two builds): by the time a
// wave reaches its V reads the copies have had the scores and the softmax to land.  The builtin stays.
__device__ __forceinline__ void glds16_asm(const void* src, char* lds_dst) {
#ifndef SS_GLDS_ASM          // default: the compiler-visible form (-DSS_GLDS_ASM: A/B builds)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
#else
    const unsigned d = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(__attribute__((address_space(3))) char*)lds_dst);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(d) : "memory");
#endif
}

// this wave's share of one tile: pieces 0..7 = K rows, 8..15 = V rows; every wave issues exactly `per` instructions so
// the vmcnt bookkeeping is uniform (surplus ones repeat piece 15: same bytes to the same place)
__device__ __forceinline__ void kv_glds(const bf16_t* base, long RS, int HD, int kv0, int T, char* stage, int wave, int nw, int per,
                                        int lane) {
    for (int i = 0; i < per; ++i) {
        int piece = wave + i * nw;
        piece = piece < 16 ? piece : 15;
        const int isv = piece >> 3;
        const int r = (piece & 7) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ (isv ? v_swz(r) : k_swz(r));
        int key = kv0 + r;
        key = key < T ? key : T - 1;
        const bf16_t* src = base + (long)key * RS + (isv + 1) * HD + c * 8;
        glds16_asm(src, stage + piece * 1024);
    }
}

// wait until at most `n` of this wave's vector-memory operations are outstanding (n is wave-uniform)
__device__ __forceinline__ void wait_vm_dyn(int n) {
    switch (n) {
        case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

template <bool DROP, bool MASK, bool DBG>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(3, 3))) void attn_fwd_bf16_kernel(AttnParams p) {
    __shared__ __attribute__((aligned(1024))) char lds[GNS * GSTAGE + (MASK ? MAXT_BIAS * 4 : 16)];
    float* kb_all = reinterpret_cast<float*>(lds + GNS * GSTAGE);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h2 = lane >> 5, ql = lane & 31;
    int qb_, bh_;
    if (!attn_block_map(p, qb_, bh_)) return;
    const int b = bh_ / p.H, h = bh_ % p.H;
    const int T = p.T;
    const long RS = 3L * p.H * 64;
    const bf16_t* base = static_cast<const bf16_t*>(p.qkv) + (long)b * T * RS + h * 64;
    const int nthr = blockDim.x, nw = nthr >> 6;
    const int per = (16 + nw - 1) / nw;
    const int q = qb_ * (nthr >> 1) + (tid >> 6) * 32 + ql;
    const int HD = p.H * 64;
    const int nt = (T + KT - 1) / KT;

    unsigned long long dbg[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long dt0 = __builtin_readcyclecounter();
    const unsigned long long dbg_start = dt0;
    bf16x8 qr[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        union { u32x4 v; bf16x8 hh; } u;
        u.v = (u32x4){0u, 0u, 0u, 0u};
        if (q < T) u.v = *reinterpret_cast<const u32x4*>(base + (long)q * RS + (2 * kk + h2) * 8);
        qr[kk] = u.hh;
    }
    kv_glds(base, RS, HD, 0, T, lds, wave, nw, per, lane);
    if (nt > 1) kv_glds(base, RS, HD, KT, T, lds + GSTAGE, wave, nw, per, lane);
    if (MASK) {
        for (int key = tid; key < nt * KT; key += nthr)
            kb_all[key] = (key < T && (!p.mask || p.mask[(long)b * T + key] != 0)) ? 0.f : NEG;
    }
    f32x16 o[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float m = NEG, lsum = 0.f;
    const int a16 = lane & 15, g16 = (lane >> 4) & 1;
    // Q must be known-landed before the loop (a later wait on it would have to be vmcnt(0) in every tile); this also waits
    // for tiles 0 and 1, which were issued in the same breath
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" ::"v"(qr[0]), "v"(qr[1]), "v"(qr[2]), "v"(qr[3]));
    dbg[0] = __builtin_readcyclecounter() - dt0;
    // per-lane fragment addresses inside a stage (the swizzle terms depend only on the lane)
    //   K: row = kb*32 + ql, chunk 2kk+h2 -> slot (2kk+h2) ^ k_swz(row); k_swz(kb*32 + ql) = k_swz(ql)
    const int krow = ql * 128, ksw = k_swz(ql);
    //   V (transposed read): row = k0 + (a16>>2) (+8), k0 = kb*32 + 16*s2 + 4*h2; chunk = db*4 + g16*2 + ((a16&3)>>1)
    //   v_swz(row) depends on bit 1 of the row = bit 1 of (a16>>2) since k0 is a multiple of 4
    const int vrow = (4 * h2 + (a16 >> 2)) * 128, vsw = v_swz(a16 >> 2);
    const int vsub = ((a16 & 3) & 1) * 8, vch = g16 * 2 + ((a16 & 3) >> 1);
    int cur = 0;
    // one 64-key tile; EDGE = the last tile (keys past T masked): peeled out of the loop so that the loop body carries a single,
    // branch-free softmax (two copies behind a branch pushed the kernel over its 168 registers)
    auto tile = [&](int it, auto edge_tag) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        const int kv0 = it * KT;
        dt0 = __builtin_readcyclecounter();
        // tile `it` landed (this wave's share; the barrier extends it to every wave's), tile it+1 may still be in flight
        if (it > 0) wait_vm_dyn(it + 1 < nt ? per : 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // stage of tile it-1 is free now (everyone is past that tile): refill it two tiles ahead
        if (it + 2 < nt) {
            int ns = cur + 2; ns = ns >= GNS ? ns - GNS : ns;
            kv_glds(base, RS, HD, kv0 + 2 * KT, T, lds + ns * GSTAGE, wave, nw, per, lane);
        }
        const char* sk = lds + cur * GSTAGE;
        const char* sv = sk + KT * 128;
        if (DBG) { unsigned long long t1 = __builtin_readcyclecounter(); dbg[4] += t1 - dt0; dt0 = t1; }

        // all eight K fragments are requested before the first MFMA (one LDS round trip per tile instead of one per
        // MFMA), and the two key blocks' accumulation chains are interleaved so no MFMA waits on its predecessor
        const bool two = !EDGE || kv0 + 32 < T;     // else the second 32-key block is all padding (its P is exactly 0)
        bf16x8 kf[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) kf[i] = ld_bf16x8(sk + (i >> 2) * 32 * 128 + krow + (((2 * (i & 3) + h2) ^ ksw) << 4));
        __builtin_amdgcn_sched_barrier(0);
        f32x16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
        if (two) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                s[0] = SS_MFMA_32x32x16(kf[kk], qr[kk], s[0], 0, 0, 0);
                s[1] = SS_MFMA_32x32x16(kf[4 + kk], qr[kk], s[1], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) s[0] = SS_MFMA_32x32x16(kf[kk], qr[kk], s[0], 0, 0, 0);
        }
        if (DBG) { asm volatile("" ::"v"(s[0][0]), "v"(s[1][15])); unsigned long long t1 = __builtin_readcyclecounter(); dbg[1] += t1 - dt0; dt0 = t1; }
        if (MASK) softmax_tile_lean<true, false>(s, kb_all + kv0, h2, p.scale_log2e, m, lsum, o, KT);
        else softmax_tile_lean<false, EDGE>(s, nullptr, h2, p.scale_log2e, m, lsum, o, T - kv0, two);
        if (DBG) { asm volatile("" ::"v"(s[0][0]), "v"(o[1][15])); unsigned long long t1 = __builtin_readcyclecounter(); dbg[2] += t1 - dt0; dt0 = t1; }
        if (DROP) {      // HF attention_probs dropout: applied to P after the normaliser is fixed
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kv0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2;
                    const unsigned idx = ((unsigned)bh_ * T + q) * T + key;                 // < 2^32: checked by the host
                    s[kb][r] = dropout_keep32(seed_fold(p.drop_seed), idx, p.drop_thresh) ? s[kb][r] * p.drop_scale : 0.f;
                }
        }
        {   // same for V: all transposed fragments first, then eight MFMAs alternating between the two d-blocks
            bf16x8 vf[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {     // i = kb*4 + s2*2 + db
                const int roff = ((i >> 2) * 32 + 16 * ((i >> 1) & 1)) * 128 + vrow;
                const int coff = ((((i & 1) * 4 + vch) ^ vsw) << 4) + vsub;
                vf[i] = tr_frag(sv + roff + coff, 8 * 128);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                if (kb == 1 && !two) break;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    bf16x8 pf;
#pragma unroll
                    for (int e = 0; e < 8; ++e) pf[e] = (bf16_t)s[kb][8 * s2 + e];
#pragma unroll
                    for (int db = 0; db < 2; ++db)
                        o[db] = SS_MFMA_32x32x16(vf[kb * 4 + s2 * 2 + db], pf, o[db], 0, 0, 0);
                }
            }
        }
        if (DBG) { asm volatile("" ::"v"(o[0][0]), "v"(o[1][15])); unsigned long long t1 = __builtin_readcyclecounter(); dbg[3] += t1 - dt0; dt0 = t1; }
        cur = cur + 1 == GNS ? 0 : cur + 1;
    };
    for (int it = 0; it < nt - 1; ++it) tile(it, std::false_type{});
    tile(nt - 1, std::true_type{});
    const float ltot = lsum + __shfl_xor(lsum, 32, 64);
    const float inv = 1.0f / ltot;
    if (q < T) {
        bf16_t* orow = static_cast<bf16_t*>(p.out) + ((long)b * T + q) * p.H * 64 + h * 64;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int d = db * 32 + 8 * r4 + 4 * h2;
                bf16x4 v = {(bf16_t)(o[db][4 * r4] * inv), (bf16_t)(o[db][4 * r4 + 1] * inv), (bf16_t)(o[db][4 * r4 + 2] * inv), (bf16_t)(o[db][4 * r4 + 3] * inv)};
                *reinterpret_cast<bf16x4*>(orow + d) = v;
            }
        if (p.lse && h2 == 0) p.lse[((long)b * p.H + h) * T + q] = m + log2f(ltot);
    }
    if (DBG && tid == 0) {
        unsigned long long* d = reinterpret_cast<unsigned long long*>(const_cast<float*>(p.delta));
        const unsigned long long tend = __builtin_readcyclecounter();
        if (qb_ == 0 && bh_ == 0) {
            for (int i = 0; i < 5; ++i) d[i] = dbg[i];
            d[5] = tend - dbg_start;
            d[6] = dbg_start;
        }
        // per-block record: start, end, HW_ID (wave/simd/cu/se), XCC_ID
        const long blin = (long)bh_ * (gridDim.x / (((p.B * p.H + 7) / 8) * 8)) + qb_;
        d[8 + blin * 4 + 0] = dbg_start;
        d[8 + blin * 4 + 1] = tend;
        d[8 + blin * 4 + 2] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
        d[8 + blin * 4 + 3] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
    }
}


// ------------------------------------------------------------------------------------------------
// 16-bit forward, long sequences without mask / dropout (round 6: the ViT towers at 384^2 / 512^2, T = 577 / 1025): 64 QUERIES PER WAVE.
// The ring kernel above gives a wave one 32-query block: every K / V fragment it reads from LDS (16 KB per 64-key tile and wave) feeds ONE
// MFMA, and with three waves per SIMD the LDS pipe is as busy as the matrix pipe.  Here a wave owns TWO 32-query blocks (a, b): each K
// fragment (MFMA A operand of S^T = K.Q^T) and each transposed V fragment (A operand of O^T = V^T.P^T) feeds two MFMAs, 32 MFMAs per tile and
// wave against the same 24 LDS reads.  Four waves (256 queries) per block, two blocks per CU (256 registers per wave).
//   * Key 0 (the class token) does not open a tile of its own: the online softmax STARTS from it - m = s_0, l = 1, O = v_0 (a rank-1 state
//     formed from one dot product per query) - and the tiles cover keys 1 .. T-1: exactly 16 full tiles at T = 1025 instead of 16 + one
//     tile holding a single key.  A remainder (T - 1) % 64 > 0 is the masked last tile.
//   * A block's waves past the last query row keep issuing their share of the K / V copies and meet the barriers, nothing else.
// Measured ratios behind the shape (tools/scratch/ubench_issue.hip, profiles/r6_ubench_issue.txt): one wave issues a plain VALU
// instruction every 5.8-6.8 cycles and v_exp_f32 every 9.8; only ~5 of them hide under one 32-cycle MFMA of the same wave, and a D = 64
// tile needs ~9 per MFMA - the softmax of one wave must run under the MFMAs of the SIMD's other wave.
// ------------------------------------------------------------------------------------------------
constexpr int W64_MINT = 512;          // (shorter sequences - 325 tokens at 288^2 - fill two 256-row blocks badly and stay on the ring kernel)
// A half-row's probability sum over a key block above this (or not finite) re-centres the row.  P and the sums have the relative precision
// of their formats at any magnitude, so the bound only has to keep P representable: 2^14 for fp16 (largest finite 65504), 2^30 for bf16.
// (At 2^8 peaked rows - a few keys far above the class token's score, the first offset - sent a wave through the re-centring path every
// few tiles: unscaled N(0,1) operands, scores of +-30 in the exponent, ran 20 % slower than scaled ones.)
#ifdef SS_HALF
constexpr float W64_BIG = 16384.0f;
#else
constexpr float W64_BIG = 1073741824.0f;
#endif

// The w64 kernel's K / V copies: buffer loads straight into LDS.  One descriptor per (batch, head) whose range ends with the batch's last
// row - rows past T read as zeros, no per-row clamp - a lane's share of the address is two 32-bit offsets fixed for the whole kernel (its row
// within an 8-row piece, its swizzled chunk, the K / V plane), the tile and the piece are a scalar offset: no 64-bit address arithmetic and
// no pointer registers in the tile loop (the ring kernel spends ~10 VALU instructions per piece on them).  Wave w copies pieces w and w + 4
// of K and of V (8 rows of 128 B each).
__device__ __forceinline__ void w64_dma(__amdgpu_buffer_rsrc_t rsrc, int offK, int offV, int row0, int rs_bytes, char* stage, int wave) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int p8 = wave + 4 * i;
        const int so = (row0 + p8 * 8) * rs_bytes;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(stage + p8 * 1024), 16, offK, so, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(stage + KT * 128 + p8 * 1024), 16, offV, so, 0, 0);
    }
}

// The softmax of one 32-key block of the w64 kernel, all of the wave's query blocks, in 4 * NQB steps of four scores (one step per MFMA of
// the phase it shares with the other key block's matrix work).  d[qb][r] IS the exponent: the score chains start from the tuple `negm`
// (= -m in all 16 registers: the MFMA's C operand) and the queries carry scale * log2(e), so P = exp2(d) with no multiply, subtract or
// maximum on the way - one v_exp, one add and half a convert per score, each exponent register dead after its v_exp.  The running offset m
// only has to keep P in range (any per-row offset cancels in O / l).  acc: four partial half-row sums per query block; the caller tests
// their total.  pk[qb][s2]: P as the PV products' B operands, the eight keys of 16-key step s2.
// QS = false (the generic entry point: q as the projection delivers it): d is the RAW score minus the offset, in raw units, and the scale
// c = scale * log2(e) goes into one multiply per score in front of the v_exp - no second rounding of q.
template <int NQB, bool QS>
__device__ __forceinline__ void softmax_w64_step(int i, const f32x16 (&d)[NQB], bf16x8 (&pk)[NQB][2], float (&acc)[NQB][4], float c) {
    constexpr int RS = 4 / NQB;                  // scores per query block and step
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
#pragma unroll
        for (int j = 0; j < RS; ++j) {
            const int r = i * RS + j;
            const float pr = __builtin_amdgcn_exp2f(QS ? d[qb][r] : d[qb][r] * c);
            acc[qb][r & 3] += pr;
            pk[qb][r >> 3][r & 7] = (bf16_t)pr;
        }
    }
    // (the sums and the packed pairs are pinned to THIS step: left alone the compiler pairs the adds of both query blocks into v_pk_add_f32
    //  and sinks all of them, and the converts, behind the phase's last MFMA - ~50 instructions that nothing covers)
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
        union { bf16x8 v; unsigned w[4]; } u;
        const int r0 = i * RS;
        u.v = pk[qb][r0 >> 3];
        if (RS == 2) asm volatile("" : "+v"(acc[qb][r0 & 3]), "+v"(acc[qb][(r0 + 1) & 3]), "+v"(u.w[(r0 & 7) >> 1]));
        else asm volatile("" : "+v"(acc[qb][0]), "+v"(acc[qb][1]), "+v"(acc[qb][2]), "+v"(acc[qb][3]), "+v"(u.w[(r0 & 7) >> 1]), "+v"(u.w[((r0 & 7) >> 1) + 1]));
        pk[qb][r0 >> 3] = u.v;
    }
}

// The rare path for one query block: some half-row's sum over a key block left [0, W64_BIG] (or is not finite) - never while m is the row's
// true maximum, since the sum of 16 probabilities is then at most 16.  d: the block's exponents, formed AGAIN by the caller (the fast path
// lets them die).  Raises m by max(0, largest exponent) per row (rows in range keep theirs), rescales O and l, rewrites the offset tuple
// and P; returns the new half-row sum and, in `inc`, what the caller still has to subtract from exponents formed with the old offset.
// (m, d, inc in the exponent's own units - raw score units with c = scale * log2(e) when q is not pre-scaled, else c = 1)
__device__ __forceinline__ float w64_recenter(const f32x16& d, bf16x8 (&pk)[2], float& m, f32x16& negm, float& lsum, f32x16 (&o)[2], float& inc,
                                              float c) {
    float mx[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 16; ++r) mx[r & 3] = fmaxf(mx[r & 3], d[r]);
    inc = fmaxf(fmaxf(mx[0], mx[1]), fmaxf(mx[2], mx[3]));
    inc = fmaxf(inc, __shfl_xor(inc, 32, 64));
    const float alpha = __builtin_amdgcn_exp2f(-inc * c);
    m += inc;
    lsum *= alpha;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
#pragma unroll
    for (int r = 0; r < 16; ++r) negm[r] = -m;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float pr = __builtin_amdgcn_exp2f((d[r] - inc) * c);
        acc[r & 3] += pr;
        pk[r >> 3][r & 7] = (bf16_t)pr;
    }
    return (acc[0] + acc[1]) + (acc[2] + acc[3]);
}

// NQB: 32-query blocks per wave - 2 = the form described above (256 registers, two waves per SIMD); 1 = the same schedule with one block
// per wave (four waves = 128 queries per block, 168 registers, three waves per SIMD: every fragment feeds one MFMA again, but a third wave
// issues beside the two - one wave issues a VALU instruction every 6-10 cycles whatever its neighbours do).
// HAS_EDGE: (T - 1) % 64 != 0 - the instantiation that carries the masked last tile (a second copy of the tile body: spilled registers;
// the towers' long sequences - 577, 1025, 2305 tokens - do not need it)
// SPLIT: the launch for the rows behind a (batch, head)'s last full block (T = 1025: ONE row).  A block then holds two row groups with TWO
// waves each: the second wave takes the tiles' second 32-key block (the first wave their first), each with the plain sequence scores ->
// softmax -> PV on its half, and the two partial softmax states (m, l, O) are merged through LDS at the end.  (One barrier per tile ties a
// block's waves to the same tile, so keys can only be split inside a tile.)  Run with one query block per wave at three blocks per CU: as
// part of the main launch the underfilled fifth block of every (batch, head) - one wave with one valid row holding a two-per-CU block slot
// for a full pass over the keys - took 21 % of the kernel's block-slot time at T = 1025.
// One block's work: (batch, head) bh_, block qblk of the rows from qbase on.  gx: blocks per (batch, head) of the launch (DBG records only).
template <int NQB, bool SPLIT, bool HAS_EDGE, bool QS, bool DBG>
__device__ __forceinline__ void w64_block(const AttnParams& p, char* lds, int bh_, int qblk, int qbase, int gx) {
    constexpr int RW = 32 * NQB, RB = (SPLIT ? 2 : 4) * RW;       // query rows per wave / per block
    constexpr int NS = 4 * NQB;                     // MFMAs per phase
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h2 = lane >> 5, ql = lane & 31;
    const int T = p.T;
    const int b = bh_ / p.H, h = bh_ % p.H;
    const long RS = 3L * p.H * 64;
    const int HD = p.H * 64;
    const bf16_t* base = static_cast<const bf16_t*>(p.qkv) + (long)b * T * RS + h * 64;
    const int grp = SPLIT ? (wave & 1) : wave, part = SPLIT ? (wave >> 1) : 0;
    const int q0 = qbase + qblk * RB + grp * RW;
    const bool active = q0 < T;                     // wave-uniform
    const int rem = (T - 1) & 63;                   // keys of the last, partial tile (tiles start at key 1)
    const int nt = ((T - 1) >> 6) + (rem ? 1 : 0);
    constexpr int per = 4;
    unsigned long long dbg[7] = {0, 0, 0, 0, 0, 0, 0}, dt0 = 0;      // DBG: {prologue, copies issued, phases 1 + 2, phase 3, phase 4, tile wait, barrier} cycle sums
    const unsigned long long dbg_start = DBG ? __builtin_readcyclecounter() : 0;
    auto stamp = [&](int slot, float touch) {
        if constexpr (DBG) {
            float t_;
            asm volatile("v_mov_b32 %0, %1" : "=v"(t_) : "v"(touch));
            const unsigned long long t1 = __builtin_readcyclecounter();
            dbg[slot] += t1 - dt0;
            dt0 = t1;
        }
    };
    if (DBG) dt0 = dbg_start;

    // Q, k_0, v_0 requested first, the copies of tiles 0 and 1 right behind them: the requests complete in order, so the prologue's arithmetic
    // (offset tuple, O = v_0) waits for its own operands only and runs while the tiles land
    const int rs_bytes = (int)RS * 2;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(base), 0, T * rs_bytes - h * 128, 0x00020000);
    const int drow = lane >> 3;                     // row within an 8-row piece; k_swz / v_swz of the row depend on the lane only
    const int offK = drow * rs_bytes + HD * 2 + (((lane & 7) ^ ((lane >> 4) & 3) ^ ((wave & 1) << 2)) << 4);
    const int offV = drow * rs_bytes + HD * 4 + (((lane & 7) ^ (((lane >> 4) & 1) << 2)) << 4);
    bf16x8 qr[NQB][4];                              // QS: the projection's q columns carry scale * log2(e) (simseg_attention_fwd_qscaled)
    const float cexp = QS ? 1.0f : p.scale_log2e;   // what an exponent is multiplied by in front of the v_exp
    f32x16 o[NQB][2];                               // [query block][d block], unnormalised
    f32x16 negm[NQB];
    float m[NQB], lsum[NQB];
    {
        bf16x8 k0[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) k0[kk] = ld_bf16x8(base + HD + (2 * kk + h2) * 8);
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) {
            int q = q0 + qb * 32 + ql;
            q = q < T ? q : T - 1;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) qr[qb][kk] = ld_bf16x8(base + (long)q * RS + (2 * kk + h2) * 8);
        }
        bf16x4 v0[2][4];
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) v0[db][r4] = *reinterpret_cast<const bf16x4*>(base + 2 * HD + db * 32 + 8 * r4 + 4 * h2);
        if (nt > 0) w64_dma(rsrc, offK, offV, 1, rs_bytes, lds, wave);
        if (nt > 1) w64_dma(rsrc, offK, offV, 1 + KT, rs_bytes, lds + GSTAGE, wave);
        // key 0 opens the softmax: m = s_0, p_0 = 1, l = 1, O = v_0.  The offset tuple -s_0 comes out of the matrix pipe: an A fragment whose 32
        // rows all hold -k_0 gives every register of the result this lane's -q.k_0 (the dot product in VALU instructions took ~300 of them
        // per wave and 2 k cycles of every block's prologue)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            union { bf16x8 v; unsigned w[4]; } u;
            u.v = k0[kk];
#pragma unroll
            for (int j = 0; j < 4; ++j) u.w[j] ^= 0x80008000u;
            k0[kk] = u.v;
        }
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) negm[qb][r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) negm[qb] = SS_MFMA_32x32x16(k0[kk], qr[qb][kk], negm[qb], 0, 0, 0);
            m[qb] = -negm[qb][0];
            lsum[qb] = (h2 == 0 && part == 0) ? 1.f : 0.f;   // (a key-split's later parts: the same offset and nothing summed yet); the two
                                                             // half-waves' sums are added at the end
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[qb][db][r] = part == 0 ? (float)v0[db][r >> 2][r & 3] : 0.f;
        }
    }
    stamp(0, o[NQB - 1][1][15]);
    const int a16 = lane & 15, g16 = (lane >> 4) & 1;
    const int krow = ql * 128, ksw = k_swz(ql);
    const int vrow = (4 * h2 + (a16 >> 2)) * 128, vsw = v_swz(a16 >> 2);
    const int vsub = ((a16 & 3) & 1) * 8, vch = g16 * 2 + ((a16 & 3) >> 1);
    int cur = 0;
    auto tile = [&](int it, auto edge_tag) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        wait_vm_dyn(it + 1 < nt ? per : 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        stamp(5, 0.f);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        stamp(6, 0.f);
        if (it + 2 < nt) {
            int ns = cur + 2; ns = ns >= GNS ? ns - GNS : ns;
            w64_dma(rsrc, offK, offV, 1 + (it + 2) * KT, rs_bytes, lds + ns * GSTAGE, wave);
        }
        stamp(1, 0.f);
        if (active) {
            const char* sk = lds + cur * GSTAGE;
            const char* sv = sk + KT * 128;
            // the score chains' C operand: -m; in the last, partial tile -1e30 for the keys past T (their K / V rows read as zeros: the
            // exponent stays -1e30, P = 0 exactly, and neither the sums nor the re-centring need a mask of their own)
            auto chain0 = [&](int qb, int kb) -> f32x16 {
                if constexpr (!EDGE) return negm[qb];
                f32x16 c;
#pragma unroll
                for (int r = 0; r < 16; ++r) c[r] = (kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2) < rem ? negm[qb][r] : NEG;
                return c;
            };
            // Per tile and wave, four phases of 4 * NQB MFMAs; the exponentials of one key block run beside the matrix work of the other:
            //   1  S(kb 0) = K0.Q'^T - m                        2  S(kb 1)  ||  P(kb 0) = exp2(S(kb 0))
            //   3  O += V0^T.P(kb 0)  ||  P(kb 1)                4  O += V1^T.P(kb 1)
            // Every K / V fragment is read from LDS once and feeds all of the wave's query blocks.
            auto kfrag = [&](int kb, int kk) { return ld_bf16x8(sk + kb * 32 * 128 + krow + (((2 * kk + h2) ^ ksw) << 4)); };
            auto vfrag = [&](int kb, int s2, int db) {
                const int roff = (kb * 32 + 16 * s2) * 128 + vrow;
                const int coff = (((db * 4 + vch) ^ vsw) << 4) + vsub;
                return tr_frag(sv + roff + coff, 8 * 128);
            };
            // MFMA i of a phase: scores - chain step kk = i / NQB of query block i % NQB; PV - (s2, db, qb) = (i / 2 NQB, i / NQB % 2, i % NQB)
            auto qk_step = [&](int i, int kb, const bf16x8 (&kf)[4], f32x16 (&s)[NQB]) {
                const int kk = i / NQB, qb = i % NQB;
                if (kk == 0) s[qb] = SS_MFMA_32x32x16(kf[0], qr[qb][0], chain0(qb, kb), 0, 0, 0);
                else s[qb] = SS_MFMA_32x32x16(kf[kk], qr[qb][kk], s[qb], 0, 0, 0);
            };
            auto pv_step = [&](int i, const bf16x8 (&vf)[4], const bf16x8 (&pk)[NQB][2]) {
                const int s2 = i / (2 * NQB), db = (i / NQB) & 1, qb = i % NQB;
                o[qb][db] = SS_MFMA_32x32x16(vf[s2 * 2 + db], pk[qb][s2], o[qb][db], 0, 0, 0);
            };
            auto kload = [&](int kb, bf16x8 (&kf)[4]) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) kf[kk] = kfrag(kb, kk);
            };
            auto vload = [&](int kb, bf16x8 (&vf)[4]) {
#pragma unroll
                for (int i = 0; i < 4; ++i) vf[i] = vfrag(kb, i >> 1, i & 1);
            };
            // the re-centring path of key block kb (wave-uniform, rare): exponents formed again; `other` = the other key block's exponents
            // when they were formed with the old offset and are still to be used
            auto recentre = [&](int kb, bf16x8 (&pk)[NQB][2], float (&ps)[NQB], f32x16 (*other)[NQB]) {
                f32x16 s[NQB];
                bf16x8 kf[4];
                kload(kb, kf);
#pragma unroll
                for (int i = 0; i < NS; ++i) qk_step(i, kb, kf, s);
#pragma unroll
                for (int qb = 0; qb < NQB; ++qb) {
                    float inc;
                    ps[qb] = w64_recenter(s[qb], pk[qb], m[qb], negm[qb], lsum[qb], o[qb], inc, cexp);
                    if (other)
#pragma unroll
                        for (int r = 0; r < 16; ++r) (*other)[qb][r] -= inc;
                }
            };
            auto in_range = [&](const float (&ps)[NQB]) {
                bool ok = ps[0] <= W64_BIG;
#pragma unroll
                for (int qb = 1; qb < NQB; ++qb) ok = ok && ps[qb] <= W64_BIG;
                return __all(ok);
            };
            f32x16 s0[NQB], s1[NQB];
            bf16x8 pk0[NQB][2], pk1[NQB][2];
            bf16x8 kf[4], vf[4];
            float acc[NQB][4], ps[NQB];
            kload(0, kf);
#pragma unroll
            for (int i = 0; i < NS; ++i) qk_step(i, 0, kf, s0);                  // phase 1
            kload(1, kf);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[qb][j] = 0.f;
#pragma unroll
            for (int i = 0; i < NS; ++i) {                                       // phase 2
                qk_step(i, 1, kf, s1);
                softmax_w64_step<NQB, QS>(i, s0, pk0, acc, cexp);
                __builtin_amdgcn_sched_barrier(0);
            }
            vload(0, vf);
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb) ps[qb] = (acc[qb][0] + acc[qb][1]) + (acc[qb][2] + acc[qb][3]);
            stamp(2, ps[NQB - 1]);
            if (!in_range(ps)) recentre(0, pk0, ps, &s1);
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb) lsum[qb] += ps[qb];
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[qb][j] = 0.f;
#pragma unroll
            for (int i = 0; i < NS; ++i) {                                       // phase 3
                pv_step(i, vf, pk0);
                softmax_w64_step<NQB, QS>(i, s1, pk1, acc, cexp);
                __builtin_amdgcn_sched_barrier(0);
            }
            vload(1, vf);
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb) ps[qb] = (acc[qb][0] + acc[qb][1]) + (acc[qb][2] + acc[qb][3]);
            stamp(3, ps[NQB - 1]);
            if (!in_range(ps)) recentre(1, pk1, ps, nullptr);
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb) lsum[qb] += ps[qb];
#pragma unroll
            for (int i = 0; i < NS; ++i) pv_step(i, vf, pk1);                    // phase 4
            stamp(4, o[NQB - 1][1][15]);
        }
        cur = cur + 1 == GNS ? 0 : cur + 1;
    };
    // the key-split form of a tile (underfilled last blocks): this wave's key block only - scores, softmax, PV in sequence
    auto tile_split = [&](int it, auto edge_tag) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        wait_vm_dyn(it + 1 < nt ? per : 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (it + 2 < nt) {
            int ns = cur + 2; ns = ns >= GNS ? ns - GNS : ns;
            w64_dma(rsrc, offK, offV, 1 + (it + 2) * KT, rs_bytes, lds + ns * GSTAGE, wave);
        }
        if (active)
#pragma unroll
        for (int kbi = 0; kbi < (SPLIT ? 1 : 2); ++kbi) {      // (not SPLIT - one query block per wave, tools only: both key blocks in sequence)
            const int kb = SPLIT ? part : kbi;
            const char* sk = lds + cur * GSTAGE + kb * 32 * 128;
            const char* sv = lds + cur * GSTAGE + KT * 128 + kb * 32 * 128;
            auto scores = [&](f32x16 (&s)[NQB]) {
                bf16x8 kf[4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) kf[kk] = ld_bf16x8(sk + krow + (((2 * kk + h2) ^ ksw) << 4));
#pragma unroll
                for (int qb = 0; qb < NQB; ++qb) {
                    f32x16 c = negm[qb];
                    if constexpr (EDGE) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) c[r] = (kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2) < rem ? negm[qb][r] : NEG;
                    }
                    s[qb] = SS_MFMA_32x32x16(kf[0], qr[qb][0], c, 0, 0, 0);
                }
#pragma unroll
                for (int kk = 1; kk < 4; ++kk)
#pragma unroll
                    for (int qb = 0; qb < NQB; ++qb) s[qb] = SS_MFMA_32x32x16(kf[kk], qr[qb][kk], s[qb], 0, 0, 0);
            };
            f32x16 s[NQB];
            bf16x8 pk[NQB][2], vf[4];
            float acc[NQB][4], ps[NQB];
            scores(s);
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[qb][j] = 0.f;
#pragma unroll
            for (int i = 0; i < NS; ++i) softmax_w64_step<NQB, QS>(i, s, pk, acc, cexp);
            bool ok = true;
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb) {
                ps[qb] = (acc[qb][0] + acc[qb][1]) + (acc[qb][2] + acc[qb][3]);
                ok = ok && ps[qb] <= W64_BIG;
            }
            if (!__all(ok)) {
                scores(s);
#pragma unroll
                for (int qb = 0; qb < NQB; ++qb) {
                    float inc;
                    ps[qb] = w64_recenter(s[qb], pk[qb], m[qb], negm[qb], lsum[qb], o[qb], inc, cexp);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {            // i = s2*2 + db   (behind the branch: 16 registers less across it)
                const int roff = (16 * (i >> 1)) * 128 + vrow;
                const int coff = ((((i & 1) * 4 + vch) ^ vsw) << 4) + vsub;
                vf[i] = tr_frag(sv + roff + coff, 8 * 128);
            }
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb) lsum[qb] += ps[qb];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int qb = 0; qb < NQB; ++qb) o[qb][db] = SS_MFMA_32x32x16(vf[s2 * 2 + db], pk[qb][s2], o[qb][db], 0, 0, 0);
        }
        cur = cur + 1 == GNS ? 0 : cur + 1;
    };
    const int nint = HAS_EDGE ? nt - 1 : nt;
    if constexpr (SPLIT || NQB == 1) {
        for (int it = 0; it < nint; ++it) tile_split(it, std::false_type{});
        if constexpr (HAS_EDGE) tile_split(nt - 1, std::true_type{});
    } else {
        for (int it = 0; it < nint; ++it) tile(it, std::false_type{});
        if constexpr (HAS_EDGE) tile(nt - 1, std::true_type{});
    }
    if constexpr (SPLIT) {
        // merge the two partial states: part 1's [o, m, l] per lane through the (now idle) ring, lane-linear
        float* xch = reinterpret_cast<float*>(lds) + grp * ((NQB * 34) * 64) + lane;
        __syncthreads();                            // the ring is idle
        if (part == 1 && active) {
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb) {
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) xch[((qb * 34) + db * 16 + r) * 64] = o[qb][db][r];
                xch[((qb * 34) + 32) * 64] = m[qb];
                xch[((qb * 34) + 33) * 64] = lsum[qb];
            }
        }
        __syncthreads();
        if (part != 0) return;
        if (active) {
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb) {
                const float mo = xch[((qb * 34) + 32) * 64], lo = xch[((qb * 34) + 33) * 64];
                const float mn = fmaxf(m[qb], mo);
                const float fa = __builtin_amdgcn_exp2f((m[qb] - mn) * cexp), fb = __builtin_amdgcn_exp2f((mo - mn) * cexp);
                m[qb] = mn;
                lsum[qb] = lsum[qb] * fa + lo * fb;
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[qb][db][r] = o[qb][db][r] * fa + xch[((qb * 34) + db * 16 + r) * 64] * fb;
            }
        }
    }
    if (!active) return;
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
        const int q = q0 + qb * 32 + ql;
        const float ltot = lsum[qb] + __shfl_xor(lsum[qb], 32, 64);
        const float inv = 1.0f / ltot;
        if (q < T) {
            bf16_t* orow = static_cast<bf16_t*>(p.out) + ((long)b * T + q) * HD + h * 64;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int d = db * 32 + 8 * r4 + 4 * h2;
                    bf16x4 v = {(bf16_t)(o[qb][db][4 * r4] * inv), (bf16_t)(o[qb][db][4 * r4 + 1] * inv), (bf16_t)(o[qb][db][4 * r4 + 2] * inv),
                                (bf16_t)(o[qb][db][4 * r4 + 3] * inv)};
                    *reinterpret_cast<bf16x4*>(orow + d) = v;
                }
            if (p.lse && h2 == 0) p.lse[((long)b * p.H + h) * T + q] = m[qb] * cexp + log2f(ltot);
        }
    }
    if (DBG && lane == 0) {      // tools/scratch/attn_w64_timeline.py: phase sums of block (0, 0)'s waves; {start, end, HW_ID, XCC_ID, phase sums} of every block's wave 0
        unsigned long long* d = reinterpret_cast<unsigned long long*>(const_cast<float*>(p.delta));
        const unsigned long long tend = __builtin_readcyclecounter();
        if (bh_ == 0 && qblk == 0) {
            for (int i = 0; i < 7; ++i) d[16 + wave * 8 + i] = dbg[i];
            d[16 + wave * 8 + 7] = tend - dbg_start;
        }
        if (wave == 0) {
            const long blin = (long)bh_ * gx + qblk;
            d[64 + blin * 12 + 0] = dbg_start;
            d[64 + blin * 12 + 1] = tend;
            d[64 + blin * 12 + 2] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
            d[64 + blin * 12 + 3] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
            for (int i = 0; i < 7; ++i) d[64 + blin * 12 + 4 + i] = dbg[i];
        }
    }
}

// the key-split form as a CALL: its registers are allocated apart from the main path's (inlined next to it, the second tile loop cost the
// main loop ~100 spilled registers)
template <bool HAS_EDGE, bool QS>
__device__ __forceinline__ void w64_split_block(const AttnParams& p, char* lds, int bh_, int qblk, int qbase) {
    w64_block<1, true, HAS_EDGE, QS, false>(p, lds, bh_, qblk, qbase, 0);
}

// p.gxw blocks per (batch, head), all on one XCD (attn_block_map's scheme): the first p.gxm of them full-size main blocks, the others
// key-split blocks of 64 rows for the rows behind (they run beside their head's main blocks and find its K / V in the XCD's L2; as a
// launch of their own they re-read all of K and V from HBM: 805 MB, 270 us, at 256 x 12 heads of 1025 tokens).  NQB = 1: tools only.
template <int NQB, bool HAS_EDGE, bool QS, bool DBG = false>
__global__ __launch_bounds__(256, NQB == 2 ? 2 : 3) void attn_fwd_w64_kernel(AttnParams p) {
    __shared__ __attribute__((aligned(1024))) char lds[GNS * GSTAGE];
    const int gx = p.gxw;
    const int L = blockIdx.x, slot = L >> 3;
    const int bh_ = (slot / gx) * 8 + (L & 7), qblk = slot % gx;
    if (bh_ >= p.B * p.H) return;
    if (NQB == 2 && qblk >= p.gxm) w64_split_block<HAS_EDGE, QS>(p, lds, bh_, qblk - p.gxm, p.gxm * 256);
    else w64_block<NQB, false, HAS_EDGE, QS, DBG>(p, lds, bh_, qblk, 0, gx);
}

// host side.  Full blocks of 4 x 64 query rows (plus a partial one when more than half a block is left over); the rows behind them go to
// key-split blocks of the same launch.  variant 6 (tools): one query block per wave throughout.
thread_local int g_w64_extra_lds = 0;      // tools (attention variant 8): unused dynamic LDS per block, so that ONE block fits a CU
template <int NQB, bool QS, bool DBG>
static void launch_w64_grid(const AttnParams& p, hipStream_t stream, unsigned blocks) {
    const int extra = g_w64_extra_lds;
    if (extra) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_w64_kernel<NQB, false, QS, DBG>), hipFuncAttributeMaxDynamicSharedMemorySize, extra);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_w64_kernel<NQB, true, QS, DBG>), hipFuncAttributeMaxDynamicSharedMemorySize, extra);
    }
    if ((p.T - 1) % 64) hipLaunchKernelGGL((attn_fwd_w64_kernel<NQB, true, QS, DBG>), dim3(blocks), dim3(256), extra, stream, p);
    else hipLaunchKernelGGL((attn_fwd_w64_kernel<NQB, false, QS, DBG>), dim3(blocks), dim3(256), extra, stream, p);
}

template <bool DBG>
static void launch_w64(const AttnParams& p0, hipStream_t stream, int nqb) {
    AttnParams p = p0;
    const long bh8 = ((long)(p.B * p.H + 7) / 8) * 8;
    // Small launches (the reference tool's one image per call: 12 heads x 1025 rows = 48 + 12 blocks for 256 CUs) take the one-query-block
    // form: 128-row blocks put more CUs to work and a lone block finishes sooner (T = 1025, 12 heads, bf16: 17.8 / 18.3 us against 22.2 / 22.4
    // at 1 / 2 images; from 4 images on - more than 256 of its blocks - the two-block form is ahead again: tools/scratch/attn_w64_small.py).
    // Sequences that leave a 256-row block 25-50 % full (577 tokens: 65 rows) keep it up to four rounds of blocks.
    if (nqb == 0) {
        const long blocks1 = (long)p.B * p.H * ((p.T + 127) / 128);
        const int left = p.T % 256;
        nqb = (blocks1 <= 256 || (left > 64 && left <= 128 && blocks1 <= 1024)) ? 1 : 2;
    }
    if (nqb != 2) {
        p.gxm = p.gxw = (p.T + 127) / 128;
        if (p.qscaled) launch_w64_grid<1, true, DBG>(p, stream, (unsigned)(p.gxw * bh8));
        else launch_w64_grid<1, false, false>(p, stream, (unsigned)(p.gxw * bh8));
        return;
    }
    int full = p.T / 256;
    const int left = p.T - full * 256;
    if (left > 64) ++full;              // (two key-split blocks - 65 rows at T = 577 - measured slower than one half-empty main block)
    p.gxm = full;
    p.gxw = full + ((left > 0 && left <= 64) ? 1 : 0);
    if (p.qscaled) launch_w64_grid<2, true, DBG>(p, stream, (unsigned)(p.gxw * bh8));
    else launch_w64_grid<2, false, false>(p, stream, (unsigned)(p.gxw * bh8));
}

// ------------------------------------------------------------------------------------------------
// Exact-mode forward on the bf16 matrix pipe (round 3): softmax(Q K^T) V of fp32 q / k / v with every product formed from the three
// exact bf16 pieces of its operands (simseg_split_bf16x3, planes form: hi / mid / lo of the packed qkv rows).  An fp32 product is nine
// piece products; the three smallest (<= 2^-24 |a||b|, an fp32 FMA's own rounding) are dropped: S accumulates (q_hi + q_mid + q_lo) k_hi
// + (q_hi + q_mid) k_mid + q_hi k_lo in fp32, the fp32 probabilities are split in registers and O accumulates the same six
// combinations of P and V pieces.  Six times the MFMA work of the bf16 kernel at sixteen times the fp32 MFMA rate: the kernel is
// MFMA-bound where the bf16 kernel is softmax-VALU-bound.  Structure of attn_fwd_bf16_kernel: 64-key tiles global -> LDS directly into
// a ring of three stages two tiles ahead ([K hi | K mid | K lo | V hi | V mid | V lo], 48 KiB per stage: one 8-wave block per CU),
// one barrier per tile, swapped S^T = K Q^T, lean online softmax.  Evaluation only: no mask, no dropout, no log-sum-exp.
// ------------------------------------------------------------------------------------------------
constexpr int X3_STAGE = 6 * KT * 128;
constexpr int X3_NS = 3;

struct AttnX3Params { const bf16_t* qkv3; long plane; float* out; int B, T, H; float scale_log2e; int dbg; };

__device__ __forceinline__ void kv_glds_x3(const bf16_t* base, long plane, long RS, int HD, int kv0, int T, char* stage, int wave, int nw, int per,
                                           int lane) {
    for (int i = 0; i < per; ++i) {                    // 48 pieces of 1 KiB per tile; every wave issues `per` (surplus ones repeat piece 47)
        int piece = wave + i * nw;
        piece = piece < 48 ? piece : 47;
        const int sel = piece >> 3;                    // 0..2: K hi / mid / lo, 3..5: V hi / mid / lo
        const int isv = sel >= 3, pc = isv ? sel - 3 : sel;
        const int r = (piece & 7) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ (isv ? v_swz(r) : k_swz(r));
        int key = kv0 + r;
        key = key < T ? key : T - 1;
        const bf16_t* src = base + pc * plane + (long)key * RS + (isv + 1) * HD + c * 8;
        glds16_asm(src, stage + piece * 1024);
    }
}

__global__ __launch_bounds__(512) void attn_fwd_x3_kernel(AttnX3Params p) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h2 = lane >> 5, ql = lane & 31;
    const int nw = blockDim.x >> 6, per = (48 + nw - 1) / nw;
    // block -> (head, query block of nw x 32 rows): consecutive ids go to consecutive XCDs; all blocks of a head share a residue
    const int gx = ((p.T + 31) / 32 + nw - 1) / nw;
    const int L = blockIdx.x, slot = L >> 3;
    const int bh_ = (slot / gx) * 8 + (L & 7), qb_ = slot % gx;
    if (bh_ >= p.B * p.H) return;
    const int b = bh_ / p.H, h = bh_ % p.H;
    const int T = p.T;
    const long RS = 3L * p.H * 64;
    const bf16_t* base = p.qkv3 + (long)b * T * RS + h * 64;
    const int q = (qb_ * nw + wave) * 32 + ql;
    const bool active = (qb_ * nw + wave) * 32 < T;         // (wave-uniform; a padding wave still copies and meets the barriers)
    const int HD = p.H * 64;
    const int nt = (T + KT - 1) / KT;
    bf16x8 qr[3][4];
#pragma unroll
    for (int pc = 0; pc < 3; ++pc)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            union { u32x4 v; bf16x8 hh; } u;
            u.v = *reinterpret_cast<const u32x4*>(base + pc * p.plane + (long)min(q, T - 1) * RS + (2 * kk + h2) * 8);     // (queries >= T: row T-1, never stored)
            qr[pc][kk] = u.hh;
        }
    kv_glds_x3(base, p.plane, RS, HD, 0, T, lds, wave, nw, per, lane);
    if (nt > 1) kv_glds_x3(base, p.plane, RS, HD, KT, T, lds + X3_STAGE, wave, nw, per, lane);
    f32x16 o[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float m = NEG, lsum = 0.f;
    const int a16 = lane & 15, g16 = (lane >> 4) & 1;
    const int krow = ql * 128, ksw = k_swz(ql);
    const int vrow = (4 * h2 + (a16 >> 2)) * 128, vsw = v_swz(a16 >> 2);
    const int vsub = ((a16 & 3) & 1) * 8, vch = g16 * 2 + ((a16 & 3) >> 1);
    // S^T of one tile: K_pc times the Q pieces whose product with it is kept (lo with hi only, mid with hi / mid, hi with all three),
    // smallest terms first
    auto scores = [&](const char* sk, f32x16 (&s)[2]) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
        for (int pc = 2; pc >= 0; --pc) {
            bf16x8 kf[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) kf[i] = ld_bf16x8(sk + pc * (KT * 128) + (i >> 2) * 32 * 128 + krow + (((2 * (i & 3) + h2) ^ ksw) << 4));
#pragma unroll
            for (int qp = 2; qp >= 0; --qp) {
                if (qp + pc > 2) continue;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    s[0] = SS_MFMA_32x32x16(kf[kk], qr[qp][kk], s[0], 0, 0, 0);
                    s[1] = SS_MFMA_32x32x16(kf[4 + kk], qr[qp][kk], s[1], 0, 0, 0);
                }
            }
        }
    };
    // Software pipeline inside the wave: the scores of tile it + 1 are issued BEFORE the softmax of tile it, so the matrix pipe works
    // through 48 MFMAs while the VALU forms the exponentials and the three pieces of P (every wave of the block passes the tile
    // barrier at the same time: without this the two waves of a SIMD do their MFMA phases together and their VALU phases together).
    // Tiles it and it + 1 are resident when iteration `it` starts; the stage of tile it - 1 is refilled with tile it + 2.
    f32x16 s[2], sn[2];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // Q and this wave's copies of tiles 0 / 1
    __builtin_amdgcn_s_barrier();
    if (active) scores(lds, s);
    int cur = 0;
    auto tile = [&](int it, auto edge_tag) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        const int kv0 = it * KT;
        if (it > 0 && !(p.dbg & 16)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's copies of tile it + 1 (issued one iteration ago)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        if (it + 2 < nt && !(p.dbg & 8)) {                  // every wave has left tile it - 1: its stage takes tile it + 2
            int ns = cur + 2; ns = ns >= X3_NS ? ns - X3_NS : ns;
            kv_glds_x3(base, p.plane, RS, HD, kv0 + 2 * KT, T, lds + ns * X3_STAGE, wave, nw, per, lane);
        }
        if (!active) { cur = cur + 1 == X3_NS ? 0 : cur + 1; return; }
        const char* sv = lds + cur * X3_STAGE + 3 * KT * 128;
        int nx = cur + 1; nx = nx == X3_NS ? 0 : nx;
        // The two waves of a SIMD (w and w + 4) run the tile's three stages in DIFFERENT orders - the first group scores(next), softmax,
        // PV; the second softmax, PV, scores(next) - so that one's exponential / piece arithmetic (VALU) runs beside the other's MFMAs:
        // every wave passes the tile barrier at the same instant, and with one order both waves of a SIMD held the matrix pipe together
        // and then the VALU together (MFMA time + VALU time per tile, measured).
        const bool first = wave < 4;
        if (!EDGE && first && !(p.dbg & 4)) scores(lds + nx * X3_STAGE, sn);
        __builtin_amdgcn_sched_barrier(0);
        if (!(p.dbg & 1)) softmax_tile_lean<false, EDGE>(s, nullptr, h2, p.scale_log2e, m, lsum, o, T - kv0);
        // the fp32 probabilities as three exact bf16 pieces (index [piece][kb * 2 + s2])
        bf16x8 pp[3][4];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float x = s[kb][8 * s2 + e];
                    const bf16_t hi = (bf16_t)x;
                    const float r1 = x - (float)hi;
                    const bf16_t mid = (bf16_t)r1;
                    pp[0][kb * 2 + s2][e] = hi;
                    pp[1][kb * 2 + s2][e] = mid;
                    pp[2][kb * 2 + s2][e] = (bf16_t)(r1 - (float)mid);
                }
        if (!(p.dbg & 2))
#pragma unroll
        for (int vc = 2; vc >= 0; --vc) {
            bf16x8 vf[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {     // i = kb*4 + s2*2 + db
                const int roff = ((i >> 2) * 32 + 16 * ((i >> 1) & 1)) * 128 + vrow;
                const int coff = ((((i & 1) * 4 + vch) ^ vsw) << 4) + vsub;
                vf[i] = tr_frag(sv + vc * (KT * 128) + roff + coff, 8 * 128);
            }
#pragma unroll
            for (int pq = 2; pq >= 0; --pq) {
                if (pq + vc > 2) continue;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                        for (int db = 0; db < 2; ++db)
                            o[db] = SS_MFMA_32x32x16(vf[kb * 4 + s2 * 2 + db], pp[pq][kb * 2 + s2], o[db], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!EDGE && !first && !(p.dbg & 4)) scores(lds + nx * X3_STAGE, sn);
        if (!EDGE) { s[0] = sn[0]; s[1] = sn[1]; }
        cur = cur + 1 == X3_NS ? 0 : cur + 1;
    };
    for (int it = 0; it < nt - 1; ++it) tile(it, std::false_type{});
    tile(nt - 1, std::true_type{});
    const float ltot = lsum + __shfl_xor(lsum, 32, 64);
    const float inv = 1.0f / ltot;
    if (q < T) {
        float* orow = p.out + ((long)b * T + q) * p.H * 64 + h * 64;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int d = db * 32 + 8 * r4 + 4 * h2;
                *reinterpret_cast<float4*>(orow + d) = make_float4(o[db][4 * r4] * inv, o[db][4 * r4 + 1] * inv, o[db][4 * r4 + 2] * inv, o[db][4 * r4 + 3] * inv);
            }
    }
}

// ------------------------------------------------------------------------------------------------
// bf16 forward, short sequences (T <= 256: ViT-B @224, every BERT caption): "resident" form.
// One block = one (batch, head); ALL of the head's K and V (T x 64 x 2 x 2 B <= 64 KiB) are copied global -> LDS once
// (global_load_lds, same swizzled stage images as the ring kernel), then every wave walks its 32-query tile over all
// key tiles with NO further barrier and no staging in the loop: the waves of a block - and the 2-4 blocks resident
// on a CU - drift apart, so the softmax VALU work of one wave runs under the MFMAs of another, and one block's load
// phase under the other blocks' compute.  (The ring kernel spent 8.3 k of its 22.5 k cycles per block staging the first
// tile and ~660 cycles per tile in wait + barrier: profiles/r1_attn_timeline.txt.)
// ------------------------------------------------------------------------------------------------
constexpr int RES_MAXT = 256;
constexpr int RES_AUTO_T = 256;          // measured (tools/dbg_attn_res.py): resident 56 vs ring 61 us at T=77, 20 vs 31 at T=25; at T=197 a tie in
                                         // isolation (214 vs 209 us) and -0.3 ms per training step (same-box A/B), so resident wherever it fits

// LDS of the resident kernels: [K rows8 x 128 B][V rows8 x 128 B][key bias], rows8 = T rounded up to 8 (rows T..rows8-1 repeat row
// T-1).  Row r of K holds its eight 16-byte chunks at slots c ^ k_swz(r), V at c ^ v_swz(r) (the images the ring kernel's fragment
// reads expect; both swizzles have a period that divides 32, so stage-relative and absolute row indices give the same slot).  The
// last, partial tile clamps its fragment rows to rows8-1 (finite values; those keys are masked / have probability exactly 0), so
// nothing is read past the matrices and a ViT-B head (T = 197) takes 51 200 B: three blocks per CU at the 1 280-byte LDS
// allocation granule (a 3 KiB guard behind V cost the third block: 202 -> 225 us).
__device__ __forceinline__ int res_rows8(int T) { return (T + 7) & ~7; }

// Effective length of sequence b for the resident kernels when the caller has dropped the padded rows from everything around the
// attention (towers.packed_text): 1 + the last unmasked key.  Keys past it are masked (their probabilities are exactly 0), and the
// caller neither reads the outputs nor uses the gradients of the query rows past it, so the kernels treat the sequence as T_eff long:
// fewer rows copied, fewer tiles.  Block-uniform; p.T when the flag is off, there is no mask, or every key is masked.
__device__ __forceinline__ int attn_teff(const AttnParams& p, int b) {
    if (p.row_start) return p.row_start[b + 1] - p.row_start[b];
    if (!p.pack || !p.mask) return p.T;
    __shared__ int sh_last[4];
    int last = -1;
    for (int j = threadIdx.x; j < p.T; j += blockDim.x)
        if (p.mask[(long)b * p.T + j] != 0) last = j;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) last = max(last, __shfl_xor(last, o, 64));
    if ((threadIdx.x & 63) == 0) sh_last[threadIdx.x >> 6] = last;
    __syncthreads();
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) last = max(last, sh_last[w]);
    __syncthreads();
    return last < 0 ? p.T : last + 1;
}

template <bool DROP, bool MASK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void attn_fwd_bf16_res_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h2 = lane >> 5, ql = lane & 31;
    const int bh_ = blockIdx.x;
    const int b = bh_ / p.H, h = bh_ % p.H;
    const int Tf = p.T, T = (MASK || p.row_start) ? attn_teff(p, b) : p.T;     // Tf: the dense row count per sequence; T: the rows this block works on
    if (T <= 0) return;                                        // (an empty sequence of a ragged batch; uniform per block)
    const long row0 = p.row_start ? (long)p.row_start[b] : (long)b * Tf;      // first row of this sequence in qkv / out
    const long RS = p.rs;
    const bf16_t* base = static_cast<const bf16_t*>(p.qkv) + row0 * RS + h * p.ps;
    const int nthr = blockDim.x, nw = nthr >> 6;
    const long HD = p.H * p.ps;                                        // from a head's q plane to its k plane
    const int nt = (T + KT - 1) / KT, q32 = (T + 31) / 32;
    const int rows8 = res_rows8(T), np = rows8 >> 3;                  // 1-KiB pieces per matrix
    char* ldsK = lds;
    char* ldsV = lds + rows8 * 128;
    float* kb_all = reinterpret_cast<float*>(ldsV + rows8 * 128);

    // K rows first (the first MFMAs need only K), then V rows; rows past T re-read row T-1
    for (int piece = (p.dbg == 102 ? 2 * np : wave); piece < 2 * np; piece += nw) {
        const int isv = piece >= np;
        const int pp = isv ? piece - np : piece;
        const int r = pp * 8 + (lane >> 3);
        const int c = (lane & 7) ^ (isv ? v_swz(r) : k_swz(r));
        const int key = r < T ? r : T - 1;
        const bf16_t* src = base + (long)key * RS + (isv + 1) * HD + c * 8;
        // (a non-temporal policy here: 0.213 -> 0.209 ms at T = 197 but 0.056 -> 0.070 ms at T = 77; left at the default)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)((isv ? ldsV : ldsK) + pp * 1024), 16, 0, 0);
    }
    // this wave's query tiles: wave, wave + nw (T <= 256 and nw = min(4, q32): at most two).  The first tile's Q rows are fetched
    // with the K / V copies; the second tile's replace them as soon as the first pass has formed its last scores.
    bf16x8 qr[4];
    auto load_q = [&](int qt) {
        const int q = qt * 32 + ql;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            union { u32x4 v; bf16x8 hh; } u;      // (queries >= T: row T-1, never stored - an unconditional request needs no wait of its own)
            // (uniform base + a 32-bit lane offset: as a 64-bit per-lane pointer the second pass's address lived in a spilled register pair)
            const unsigned voff = ((unsigned)min(q, T - 1) * (unsigned)RS + (unsigned)((2 * kk + h2) * 8)) * 2u;
            u.v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(base) + voff);
            qr[kk] = u.hh;
        }
    };
    load_q(wave);
    if (MASK) {      // after the Q requests: storing a loaded value to LDS waits for everything requested before it
        for (int key = tid; key < nt * KT; key += nthr)
            kb_all[key] = (key < T && (!p.mask || p.mask[(long)b * Tf + key] != 0)) ? 0.f : NEG;
    }
    const int a16 = lane & 15, g16 = (lane >> 4) & 1;
    const int krow = ql * 128, ksw = k_swz(ql);
    const int vrow = (4 * h2 + (a16 >> 2)) * 128, vsw = v_swz(a16 >> 2);
    const int vsub = ((a16 & 3) & 1) * 8, vch = g16 * 2 + ((a16 & 3) >> 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" ::"v"(qr[0]), "v"(qr[1]), "v"(qr[2]), "v"(qr[3]));
    __syncthreads();

#pragma unroll 1
    for (int qt = wave; qt < q32; qt += nw) {
        const int q = qt * 32 + ql;
        f32x16 o[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
        float m = NEG, lsum = 0.f;
        // one 64-key tile; EDGE = the last, partial tile (keys past T masked; its second 32-key block may be all padding)
        auto tile = [&](int it, auto edge_tag) {
            constexpr bool EDGE = decltype(edge_tag)::value;
            const int kv0 = it * KT;
            const char* sk = ldsK + kv0 * 128;
            const char* sv = ldsV + kv0 * 128;
            const bool two = !EDGE || kv0 + 32 < T;     // else the second 32-key block is all padding (its P is exactly 0)
            f32x16 s[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
            {
                bf16x8 kf[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (!EDGE) {
                        kf[i] = ld_bf16x8(sk + (i >> 2) * 32 * 128 + krow + (((2 * (i & 3) + h2) ^ ksw) << 4));
                    } else {
                        int r = kv0 + (i >> 2) * 32 + ql;
                        r = r < rows8 ? r : rows8 - 1;
                        kf[i] = ld_bf16x8(ldsK + r * 128 + (((2 * (i & 3) + h2) ^ k_swz(r)) << 4));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);      // all fragments requested before the first MFMA (one LDS round trip, not eight)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    s[0] = SS_MFMA_32x32x16(kf[kk], qr[kk], s[0], 0, 0, 0);
                    if (two) s[1] = SS_MFMA_32x32x16(kf[4 + kk], qr[kk], s[1], 0, 0, 0);
                }
            }
            if (EDGE && qt + nw < q32) load_q(qt + nw);      // Q of the next pass: lands under this tile's softmax, PV and stores
            // the transposed V fragments are requested before the softmax arithmetic: their LDS round trip runs under it
            bf16x8 vf[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {     // i = kb*4 + s2*2 + db
                if (!EDGE) {
                    const int roff = ((i >> 2) * 32 + 16 * ((i >> 1) & 1)) * 128 + vrow;
                    const int coff = ((((i & 1) * 4 + vch) ^ vsw) << 4) + vsub;
                    vf[i] = tr_frag(sv + roff + coff, 8 * 128);
                } else {
                    const int r0 = kv0 + (i >> 2) * 32 + 16 * ((i >> 1) & 1) + 4 * h2 + (a16 >> 2);
                    const int ra = r0 < rows8 ? r0 : rows8 - 1, rb = r0 + 8 < rows8 ? r0 + 8 : rows8 - 1;
                    const char* pa = ldsV + ra * 128 + ((((i & 1) * 4 + vch) ^ v_swz(ra)) << 4) + vsub;
                    const char* pb = ldsV + rb * 128 + ((((i & 1) * 4 + vch) ^ v_swz(rb)) << 4) + vsub;
                    union { struct { s16x4 lo, hi; } h; bf16x8 v; } u;
                    u.h.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(pa));
                    u.h.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(pb));
                    vf[i] = u.v;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (MASK) softmax_tile_lean<true, false>(s, kb_all + kv0, h2, p.scale_log2e, m, lsum, o, KT);
            else softmax_tile_lean<false, EDGE>(s, nullptr, h2, p.scale_log2e, m, lsum, o, T - kv0, two);
            if (DROP) {      // HF attention_probs dropout: applied to P after the normaliser is fixed
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = kv0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2;
                        const unsigned idx = ((unsigned)bh_ * Tf + q) * Tf + key;                 // < 2^32: checked by the host
                        s[kb][r] = dropout_keep32(seed_fold(p.drop_seed), idx, p.drop_thresh) ? s[kb][r] * p.drop_scale : 0.f;
                    }
            }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                if (kb == 1 && !two) break;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    bf16x8 pf;
#pragma unroll
                    for (int e = 0; e < 8; ++e) pf[e] = (bf16_t)s[kb][8 * s2 + e];
#pragma unroll
                    for (int db = 0; db < 2; ++db)
                        o[db] = SS_MFMA_32x32x16(vf[kb * 4 + s2 * 2 + db], pf, o[db], 0, 0, 0);
                }
            }
        };
        if (p.dbg != 103) {
            for (int it = 0; it < nt - 1; ++it) tile(it, std::false_type{});
            tile(nt - 1, std::true_type{});            // (a full last tile goes through the masked form too: same result)
        }
        const float ltot = lsum + __shfl_xor(lsum, 32, 64);
        const float inv = 1.0f / ltot;
        if (q < T) {
            bf16_t* orow = static_cast<bf16_t*>(p.out) + (row0 + q) * p.H * 64 + h * 64;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int d = db * 32 + 8 * r4 + 4 * h2;
                    bf16x4 v = {(bf16_t)(o[db][4 * r4] * inv), (bf16_t)(o[db][4 * r4 + 1] * inv), (bf16_t)(o[db][4 * r4 + 2] * inv), (bf16_t)(o[db][4 * r4 + 3] * inv)};
                    *reinterpret_cast<bf16x4*>(orow + d) = v;
                }
            if (p.lse && h2 == 0) p.lse[((long)b * p.H + h) * Tf + q] = m + log2f(ltot);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// bf16 forward, ViT-B training shape (128 < T <= 256, no mask, no dropout): PERSISTENT form with a loader wave (round 4).
// The resident kernel above is one block per head: copy K / V (50 KB at T = 197) -> wait -> 2 query passes -> store, three such blocks per
// CU at independent phases.  Measured (profiles/r2_attn_fwd_ablation.txt): its copies alone take 150 us, its tile loops alone 164 us, the
// kernel 214 us - a block that computes has no copies in flight and a CU's HBM share is proportional to what it has in flight, so most
// of one component is exposed.  Here ONE block per CU walks its heads: wave q32 only issues global -> LDS copies (the next head's K and V
// into the other LDS stage, all 2 * rows8 / 8 pieces at once) and waits for them; waves 0 .. q32-1 own one 32-query tile each of the
// CURRENT head - one pass instead of two, no second Q load, 7 of 8 waves busy instead of 7 of 8 passes - with the next head's Q rows
// requested into a second register set at the start of the head.  One barrier per head hands the stages over.  A head's copies have the
// whole tile loop of the previous head to land: the copy stream never stops, the matrix / VALU pipes never wait for it.
// ------------------------------------------------------------------------------------------------
constexpr int PERS_MAXT = 224;       // 7 compute waves + the loader = 8 waves = two per SIMD: the whole 256-register budget per wave

__host__ __device__ inline int pers_stage_bytes(int T) { return 2 * ((T + 7) & ~7) * 128; }

template <int DUMMY = 0>
__global__ __launch_bounds__(512) void attn_fwd_bf16_pers_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h2 = lane >> 5, ql = lane & 31;
    const int T = p.T;
    const long RS = 3L * p.H * 64;
    const int HD = p.H * 64;
    const int nt = (T + KT - 1) / KT, q32 = (T + 31) / 32;
    const int rows8 = res_rows8(T), np = rows8 >> 3;
    const int stage_bytes = 2 * rows8 * 128;
    const int nheads = p.B * p.H;
    const int mine = (nheads - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;      // heads blockIdx.x, + gridDim.x, ...
    const bool loader = wave == q32;
    const bf16_t* qkv = static_cast<const bf16_t*>(p.qkv);

    auto issue = [&](int hd, int st) {             // the loader's: every piece of head hd's K (pieces 0..np-1) and V into stage st
        const int b = hd / p.H, h = hd % p.H;
        const bf16_t* base = qkv + (long)b * T * RS + h * 64;
        char* dstK = lds + st * stage_bytes;
        char* dstV = dstK + rows8 * 128;
        for (int piece = 0; piece < 2 * np; ++piece) {
            const int isv = piece >= np;
            const int pp = isv ? piece - np : piece;
            const int r = pp * 8 + (lane >> 3);
            const int c = (lane & 7) ^ (isv ? v_swz(r) : k_swz(r));
            const int key = r < T ? r : T - 1;
            const bf16_t* src = base + (long)key * RS + (isv + 1) * HD + c * 8;
            // (from inline asm: through the builtin the compiler knows that LDS writes are in flight somewhere in this kernel and puts
            //  s_waitcnt vmcnt(0) in front of every transposed V read of the COMPUTE waves - draining their output stores and their Q
            //  requests once per tile; the loader's own wait in front of the head barrier is the only one that is needed)
            const unsigned d = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(__attribute__((address_space(3))) char*)((isv ? dstV : dstK) + pp * 1024));
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(d) : "memory");
        }
    };
    bf16x8 qr[4];
    auto load_q = [&](int hd, bf16x8 (&dst)[4]) {  // a compute wave's: its 32 query rows of head hd (rows >= T: row T-1, never stored)
        const int b = hd / p.H, h = hd % p.H;
        const bf16_t* base = qkv + (long)b * T * RS + h * 64;
        const int q = wave * 32 + ql;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            // (plain loads: the compiler then puts s_waitcnt vmcnt(0) in front of the first MFMA that reads these registers, which also
            //  drains the previous head's output stores.  Requesting them from inline asm instead - invisible to the wait counting, with
            //  the loop's counted wait covering them - produced NaNs: the compiler believes an asm output is valid at once and moved
            //  the registers before the data had landed.)
            union { u32x4 v; bf16x8 hh; } u;
            u.v = *reinterpret_cast<const u32x4*>(base + (long)min(q, T - 1) * RS + (2 * kk + h2) * 8);
            dst[kk] = u.hh;
        }
    };
    if (mine <= 0) return;
    // (p.dbg: timing ablations with WRONG results - 41 = no tile loop, 42 = no K / V copies, 43 = no output stores: tools/attn_fwd_ab.py)
    if (loader) { if (p.dbg != 42) issue(blockIdx.x, 0); }
    else load_q(blockIdx.x, qr);

    const int a16 = lane & 15, g16 = (lane >> 4) & 1;
    const int krow = ql * 128, ksw = k_swz(ql);
    const int vrow = (4 * h2 + (a16 >> 2)) * 128, vsw = v_swz(a16 >> 2);
    const int vsub = ((a16 & 3) & 1) * 8, vch = g16 * 2 + ((a16 & 3) >> 1);

#pragma unroll 1
    for (int i = 0; i < mine; ++i) {
        const int hd = blockIdx.x + i * gridDim.x;
        const int st = i & 1;
        // everything requested for head i (the loader's copies; this wave's Q rows) has landed before the barrier; behind it the other
        // stage - head i-1's - is free
        // (compute waves, from the second head on: the 4 Q requests are OLDER than the previous head's 8 output stores + 1 lse store, and
        //  vmcnt retires in order - waiting for "at most 9 / 8 outstanding" leaves the stores in flight across the barrier.  A full drain here
        //  exposed the stores' acknowledgement latency once per head: 228 vs 147 us for the layer, tools/attn_fwd_ab.py.)
        if (loader || i == 0 || p.dbg == 43) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (p.lse) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();               // (the bare barrier: __syncthreads() is a fence + barrier and drains vmcnt again)
        __builtin_amdgcn_sched_barrier(0);
        if (loader) {
            if (i + 1 < mine && p.dbg != 42) issue(hd + gridDim.x, st ^ 1);
            continue;
        }
        const char* ldsK = lds + st * stage_bytes;
        const char* ldsV = ldsK + rows8 * 128;
        const int b = hd / p.H, h = hd % p.H;
        const int q = wave * 32 + ql;
        f32x16 o[2];
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[ii][r] = 0.f;
        float m = NEG, lsum = 0.f;
        // Software pipeline over the key tiles (two score register sets): while the VALU works through tile it's softmax, the matrix pipe
        // forms tile it+1's scores from fragments requested a stage earlier - a lone wave spent a third of its cycles in s_waitcnt (LDS round
        // trips in front of the first score MFMA and behind the maximum's cross-lane exchange) and a quarter stalled at issue
        // (profiles/r2_pmc_attn_fwd_resident.txt); with two waves per SIMD there is nobody else to fill those holes.
        auto load_kf = [&](int it, bf16x8 (&kf)[8], auto edge_tag) {
            constexpr bool EDGE = decltype(edge_tag)::value;
            const int kv0 = it * KT;
            const char* sk = ldsK + kv0 * 128;
#pragma unroll
            for (int ii = 0; ii < 8; ++ii) {
                if (!EDGE) {
                    kf[ii] = ld_bf16x8(sk + (ii >> 2) * 32 * 128 + krow + (((2 * (ii & 3) + h2) ^ ksw) << 4));
                } else {
                    int r = kv0 + (ii >> 2) * 32 + ql;
                    r = r < rows8 ? r : rows8 - 1;
                    kf[ii] = ld_bf16x8(ldsK + r * 128 + (((2 * (ii & 3) + h2) ^ k_swz(r)) << 4));
                }
            }
        };
        auto load_vf = [&](int it, bf16x8 (&vf)[8], auto edge_tag) {
            constexpr bool EDGE = decltype(edge_tag)::value;
            const int kv0 = it * KT;
            const char* sv = ldsV + kv0 * 128;
#pragma unroll
            for (int ii = 0; ii < 8; ++ii) {     // ii = kb*4 + s2*2 + db
                if (!EDGE) {
                    const int roff = ((ii >> 2) * 32 + 16 * ((ii >> 1) & 1)) * 128 + vrow;
                    const int coff = ((((ii & 1) * 4 + vch) ^ vsw) << 4) + vsub;
                    vf[ii] = tr_frag(sv + roff + coff, 8 * 128);
                } else {
                    const int r0 = kv0 + (ii >> 2) * 32 + 16 * ((ii >> 1) & 1) + 4 * h2 + (a16 >> 2);
                    const int ra = r0 < rows8 ? r0 : rows8 - 1, rb = r0 + 8 < rows8 ? r0 + 8 : rows8 - 1;
                    const char* pa = ldsV + ra * 128 + ((((ii & 1) * 4 + vch) ^ v_swz(ra)) << 4) + vsub;
                    const char* pb = ldsV + rb * 128 + ((((ii & 1) * 4 + vch) ^ v_swz(rb)) << 4) + vsub;
                    union { struct { s16x4 lo, hi; } hh; bf16x8 v; } u;
                    u.hh.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(pa));
                    u.hh.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(pb));
                    vf[ii] = u.v;
                }
            }
        };
        auto scores = [&](const bf16x8 (&kf)[8], f32x16 (&s)[2]) {      // (both 32-key blocks always: a branch per MFMA would end the scheduling region;
#pragma unroll                                                          //  an all-padding block of the last tile costs four MFMAs once per head)
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                s[0] = SS_MFMA_32x32x16(kf[kk], qr[kk], s[0], 0, 0, 0);
                s[1] = SS_MFMA_32x32x16(kf[4 + kk], qr[kk], s[1], 0, 0, 0);
            }
        };
        auto pv = [&](const bf16x8 (&vf)[8], const f32x16 (&s)[2], bool two) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                if (kb == 1 && !two) break;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    bf16x8 pf;
#pragma unroll
                    for (int e = 0; e < 8; ++e) pf[e] = (bf16_t)s[kb][8 * s2 + e];
#pragma unroll
                    for (int db = 0; db < 2; ++db)
                        o[db] = SS_MFMA_32x32x16(vf[kb * 4 + s2 * 2 + db], pf, o[db], 0, 0, 0);
                }
            }
        };
        const int kvl = (nt - 1) * KT;
        const bool two_last = kvl + 32 < T;         // the last tile's second 32-key block holds keys
        f32x16 sc[2], sn[2];
        bf16x8 kf[8], vf[8];
        if (nt == 1) load_kf(0, kf, std::true_type{}); else load_kf(0, kf, std::false_type{});
        scores(kf, sc);
        if (p.dbg != 41) {
#pragma unroll 1
        for (int it = 0; it + 2 < nt; ++it) {        // tiles it and it+1 interior: tile it+1's scores are formed beside tile it's softmax
            load_vf(it, vf, std::false_type{});
            load_kf(it + 1, kf, std::false_type{});
            __builtin_amdgcn_sched_barrier(0);
            scores(kf, sn);
            softmax_tile_lean<false, false>(sc, nullptr, h2, p.scale_log2e, m, lsum, o, KT);
            pv(vf, sc, true);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) sc[kb] = sn[kb];
        }
        if (nt > 1) {                                 // tile nt-2 (interior) beside the scores of the last, partial tile
            load_vf(nt - 2, vf, std::false_type{});
            load_kf(nt - 1, kf, std::true_type{});
            __builtin_amdgcn_sched_barrier(0);
            scores(kf, sn);
            softmax_tile_lean<false, false>(sc, nullptr, h2, p.scale_log2e, m, lsum, o, KT);
            pv(vf, sc, true);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) sc[kb] = sn[kb];
        }
        }
        // the head's last scores are formed: Q of the NEXT head replaces it (lands under the last tile's softmax / PV, the stores and the barrier)
        if (i + 1 < mine) load_q(hd + gridDim.x, qr);
        load_vf(nt - 1, vf, std::true_type{});
        __builtin_amdgcn_sched_barrier(0);
        softmax_tile_lean<false, true>(sc, nullptr, h2, p.scale_log2e, m, lsum, o, T - kvl, two_last);
        pv(vf, sc, two_last);
        const float ltot = lsum + __shfl_xor(lsum, 32, 64);
        const float inv = 1.0f / ltot;
        if (q < T && p.dbg != 43) {
            bf16_t* orow = static_cast<bf16_t*>(p.out) + ((long)b * T + q) * HD + h * 64;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int d = db * 32 + 8 * r4 + 4 * h2;
                    bf16x4 v = {(bf16_t)(o[db][4 * r4] * inv), (bf16_t)(o[db][4 * r4 + 1] * inv), (bf16_t)(o[db][4 * r4 + 2] * inv), (bf16_t)(o[db][4 * r4 + 3] * inv)};
                    *reinterpret_cast<bf16x4*>(orow + d) = v;
                }
            if (p.lse && h2 == 0) p.lse[((long)b * p.H + h) * T + q] = m + log2f(ltot);
        }
    }
}

int launch_fwd_pers(const AttnParams& p, hipStream_t stream) {
    static bool configured = false;
    static int ncu = 256;
    auto kern = attn_fwd_bf16_pers_kernel<0>;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * pers_stage_bytes(PERS_MAXT));
        if (e != hipSuccess) return simseg_set_error("attention_fwd: cannot reserve LDS: %s", hipGetErrorString(e));
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount;
        configured = true;
    }
    const int q32 = (p.T + 31) / 32;
    const int nheads = p.B * p.H;
    hipLaunchKernelGGL(kern, dim3((unsigned)(nheads < ncu ? nheads : ncu)), dim3((q32 + 1) * 64), 2 * pers_stage_bytes(p.T), stream, p);
    return 0;
}

__host__ int res_smem(int T, bool mask) { return 2 * ((T + 7) & ~7) * 128 + (mask ? ((T + KT - 1) / KT) * KT * 4 : 0); }

template <bool DROP, bool MASK>
int launch_fwd_res(const AttnParams& p, hipStream_t stream) {
    const int q32 = (p.T + 31) / 32;
    static bool configured = false;
    auto kern = attn_fwd_bf16_res_kernel<DROP, MASK>;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, res_smem(RES_MAXT, true));
        if (e != hipSuccess) return simseg_set_error("attention_fwd: cannot reserve LDS: %s", hipGetErrorString(e));
        configured = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(p.B * p.H)), dim3((q32 < 4 ? q32 : 4) * 64), res_smem(p.T, MASK), stream, p);
    return 0;
}

// ------------------------------------------------------------------------------------------------
// bf16 backward.  delta[b,h,q] = sum_d dO.O ;  P = exp2(S*c + bias - lse) ;  dS = P o (dP - delta) * scale
//   kernel A (per 128-key tile): dK = dS^T.Q, dV = P_d^T.dO      (queries streamed through LDS, 32 at a time)
//   kernel B (per 128-query tile): dQ = dS.K                     (keys streamed through LDS, 64 at a time)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout,
                                                         float* __restrict__ delta, int B, int T, int H) {
    // one 8-lane group per (b, t, h): 64 d = 8 lanes x 8 elements
    const long gid = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const int sub = threadIdx.x & 7;
    if (gid >= (long)B * T * H) return;
    const long off = gid * 64 + sub * 8;
    const bf16x8 a = ld_bf16x8(o + off), g = ld_bf16x8(dout + off);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s += (float)a[e] * (float)g[e];
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
    if (sub == 0) {
        const int h = (int)(gid % H);
        const long bt = gid / H;
        const int t = (int)(bt % T);
        const int b = (int)(bt / T);
        delta[((long)b * H + h) * T + t] = s;
    }
}

constexpr int QT = 32;   // queries per LDS tile in the dK/dV kernel
constexpr int QP = 144;  // pitch of the Q / dO tiles (read both k-contiguous and transposed)
constexpr int QBUF = 2 * QT * QP + 2 * QT * 4;   // one stage: Q tile, dO tile, lse[32], delta[32]

__device__ __forceinline__ void store_bf16x4(bf16_t* dst, float a, float b, float c, float d) {
    const bf16x4 v = {(bf16_t)a, (bf16_t)b, (bf16_t)c, (bf16_t)d};
    *reinterpret_cast<bf16x4*>(dst) = v;
}

// Schedule notes (both backward kernels, same as the forward): every LDS fragment a group of MFMAs needs is requested
// before the first MFMA of the group (one LDS round trip per group, not one per MFMA); the two accumulators of a group
// alternate so no MFMA waits on its predecessor; the gradient products are formed TRANSPOSED (A = transposed LDS
// fragment, B = the probabilities), which leaves each lane with four consecutive d of its own key / query row: 8-byte
// stores instead of 2-byte ones.  The Q/dO stage is double buffered: one barrier per 32-query tile.
template <bool DROP>
__global__ __launch_bounds__(512) void attn_bwd_dkv_kernel(AttnParams p) {
    __shared__ __attribute__((aligned(16))) char lds[2 * QBUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h2 = lane >> 5, kl = lane & 31;
    int qb_, bh_;
    if (!attn_block_map(p, qb_, bh_)) return;
    const int b = bh_ / p.H, h = bh_ % p.H;
    const int T = p.T;
    const long RS = 3L * p.H * 64, OS = (long)p.H * 64;
    const bf16_t* base = static_cast<const bf16_t*>(p.qkv) + (long)b * T * RS + h * 64;
    const bf16_t* gbase = static_cast<const bf16_t*>(p.dout) + (long)b * T * OS + h * 64;
    const float* lse_g = p.lse + ((long)b * p.H + h) * T;
    const float* del_g = p.delta + ((long)b * p.H + h) * T;
    const int nthr = blockDim.x;
    const int key = qb_ * (nthr >> 1) + wave * 32 + kl;
    const bool kvalid = key < T;
    const float kb_ = (kvalid && (!p.mask || p.mask[(long)b * T + key] != 0)) ? 0.f : NEG;

    bf16x8 kr[4], vr[4];     // this lane's K and V row chunks: B operands of S = Q.K^T and dP = dO.V^T
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        union { u32x4 v; bf16x8 hh; } uk, uv;
        uk.v = (u32x4){0u, 0u, 0u, 0u}; uv.v = uk.v;
        if (kvalid) {
            uk.v = *reinterpret_cast<const u32x4*>(base + (long)key * RS + p.H * 64 + (2 * kk + h2) * 8);
            uv.v = *reinterpret_cast<const u32x4*>(base + (long)key * RS + 2 * p.H * 64 + (2 * kk + h2) * 8);
        }
        kr[kk] = uk.hh; vr[kk] = uv.hh;
    }
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[i][r] = 0.f; dv[i][r] = 0.f; }
    const int a16 = lane & 15, g16 = (lane >> 4) & 1;
    const float scale = p.scale_log2e * 0.6931471805599453f;

    // Q / dO tiles of 32 queries: fetched into registers during the previous tile's MFMAs, written to the other stage.
    // 256 16-byte pieces per operand tile; a block has >= 1 wave, so up to 4 pieces per thread (2 when nthr >= 128).
    const int qq_ = tid >> 3, c_ = tid & 7;              // piece i of this thread: query qq_ + i * (nthr / 8), chunk c_
    const int qstep = nthr >> 3;
    u32x4 nq[2], ng[2];
    float nl = 1e30f, nd = 0.f;
    auto qg_fetch = [&](int q0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int qq = qq_ + i * qstep;
            u32x4 qv = {0u, 0u, 0u, 0u}, gv = qv;
            if (qq < QT && q0 + qq < T) {
                qv = *reinterpret_cast<const u32x4*>(base + (long)(q0 + qq) * RS + c_ * 8);
                gv = *reinterpret_cast<const u32x4*>(gbase + (long)(q0 + qq) * OS + c_ * 8);
            }
            nq[i] = qv; ng[i] = gv;
        }
        if (tid < QT) {
            const bool v = q0 + tid < T;
            nl = v ? lse_g[q0 + tid] : 1e30f;            // -> P = 0 for padded queries
            nd = v ? del_g[q0 + tid] : 0.f;
        }
    };
    auto qg_commit = [&](char* st) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int qq = qq_ + i * qstep;
            if (qq < QT) {
                *reinterpret_cast<u32x4*>(st + qq * QP + c_ * 16) = nq[i];
                *reinterpret_cast<u32x4*>(st + QT * QP + qq * QP + c_ * 16) = ng[i];
            }
        }
        if (tid < QT) {
            float* f = reinterpret_cast<float*>(st + 2 * QT * QP);
            f[tid] = nl; f[QT + tid] = nd;
        }
    };
    if (nthr >= 128) {
        qg_fetch(0);
        qg_commit(lds);
    } else {                                              // single-wave block (T <= 32): four pieces per thread
        for (int idx = tid; idx < QT * 8; idx += nthr) {
            const int qq = idx >> 3, c = idx & 7;
            u32x4 qv = {0u, 0u, 0u, 0u}, gv = qv;
            if (qq < T) {
                qv = *reinterpret_cast<const u32x4*>(base + (long)qq * RS + c * 8);
                gv = *reinterpret_cast<const u32x4*>(gbase + (long)qq * OS + c * 8);
            }
            *reinterpret_cast<u32x4*>(lds + qq * QP + c * 16) = qv;
            *reinterpret_cast<u32x4*>(lds + QT * QP + qq * QP + c * 16) = gv;
        }
        if (tid < QT) {
            float* f = reinterpret_cast<float*>(lds + 2 * QT * QP);
            f[tid] = tid < T ? lse_g[tid] : 1e30f;
            f[QT + tid] = tid < T ? del_g[tid] : 0.f;
        }
    }
    __syncthreads();
    int cur = 0;
    for (int q0 = 0; q0 < T; q0 += QT) {
        const bool more = q0 + QT < T;                    // more than one tile implies T > 32, i.e. nthr >= 128
        if (more) qg_fetch(q0 + QT);
        const char* cq = lds + cur * QBUF;
        const char* cg = cq + QT * QP;
        const float* cl = reinterpret_cast<const float*>(cq + 2 * QT * QP);
        // S[q][key], dP[q][key]: MFMA rows = queries (A from LDS), columns = keys (B = this lane's K / V row)
        bf16x8 aq[4], ag[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            aq[kk] = ld_bf16x8(cq + kl * QP + (2 * kk + h2) * 16);
            ag[kk] = ld_bf16x8(cg + kl * QP + (2 * kk + h2) * 16);
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            s = SS_MFMA_32x32x16(aq[kk], kr[kk], s, 0, 0, 0);
            dp = SS_MFMA_32x32x16(ag[kk], vr[kk], dp, 0, 0, 0);
        }
        // requested while the MFMAs run: lse / delta of this lane's 16 query rows, and the transposed dO / Q fragments
        float4 l4[4], d4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            l4[j] = *reinterpret_cast<const float4*>(cl + 8 * j + 4 * h2);
            d4[j] = *reinterpret_cast<const float4*>(cl + QT + 8 * j + 4 * h2);
        }
        bf16x8 gf[4], qf[4];                              // index s2 * 2 + db
        auto load_tr = [&]() {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int off = (16 * (i >> 1) + 4 * h2 + (a16 >> 2)) * QP + ((i & 1) * 32 + 16 * g16 + 4 * (a16 & 3)) * 2;
                gf[i] = tr_frag(cg + off, 8 * QP);        // dO^T fragment: [d][q-slots]
                qf[i] = tr_frag(cq + off, 8 * QP);        // Q^T fragment
            }
        };
        if (!DROP) load_tr();                             // (the dropout variant has no registers to spare for this overlap)
        __builtin_amdgcn_sched_barrier(0);
        // lane: key column kl, rows q = (r%4) + 8*(r/4) + 4*h2
        float pd[16], ds[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float lq = reinterpret_cast<const float*>(&l4[r >> 2])[r & 3];
            const float dq_ = reinterpret_cast<const float*>(&d4[r >> 2])[r & 3];
            const float pr = __builtin_amdgcn_exp2f(fminf(s[r] * p.scale_log2e + kb_ - lq, 0.f));
            float keep = 1.f;
            if (DROP) {
                const int qq = (r & 3) + 8 * (r >> 2) + 4 * h2;
                const unsigned idx = ((unsigned)bh_ * T + (q0 + qq)) * T + key;
                keep = dropout_keep32(seed_fold(p.drop_seed), idx, p.drop_thresh) ? p.drop_scale : 0.f;
            }
            pd[r] = pr * keep;                                   // dropped probabilities feed dV
            ds[r] = pr * (dp[r] * keep - dq_) * scale;          // dS feeds dK
        }
        // dV^T[d][key] += dO^T[d][q] . P[q][key] ;  dK^T[d][key] += Q^T[d][q] . dS[q][key]
        if (DROP) {
            __builtin_amdgcn_sched_barrier(0);
            load_tr();
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            bf16x8 pf, df;
#pragma unroll
            for (int e = 0; e < 8; ++e) { pf[e] = (bf16_t)pd[8 * s2 + e]; df[e] = (bf16_t)ds[8 * s2 + e]; }
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                dv[db] = SS_MFMA_32x32x16(gf[s2 * 2 + db], pf, dv[db], 0, 0, 0);
                dk[db] = SS_MFMA_32x32x16(qf[s2 * 2 + db], df, dk[db], 0, 0, 0);
            }
        }
        if (more) {
            qg_commit(lds + (cur ^ 1) * QBUF);            // the other stage: last read before the previous barrier
            __syncthreads();
            cur ^= 1;
        }
    }
    // dk/dv accumulators (transposed): column = this lane's key, rows d = db*32 + (r%4) + 8*(r/4) + 4*h2
    if (kvalid) {
        bf16_t* drow = static_cast<bf16_t*>(p.dqkv) + (long)b * T * RS + h * 64 + (long)key * RS;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int d = db * 32 + 8 * r4 + 4 * h2;
                store_bf16x4(drow + p.H * 64 + d, dk[db][4 * r4], dk[db][4 * r4 + 1], dk[db][4 * r4 + 2], dk[db][4 * r4 + 3]);
                store_bf16x4(drow + 2 * p.H * 64 + d, dv[db][4 * r4], dv[db][4 * r4 + 1], dv[db][4 * r4 + 2], dv[db][4 * r4 + 3]);
            }
    }
}

template <bool DROP>
__global__ __launch_bounds__(512) void attn_bwd_dq_kernel(AttnParams p) {
    __shared__ __attribute__((aligned(16))) char lds[KT * KP16 + KT * KP16 + KT * 4];
    char* ldsK = lds;
    char* ldsV = lds + KT * KP16;
    float* kbias = reinterpret_cast<float*>(lds + 2 * KT * KP16);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h2 = lane >> 5, ql = lane & 31;
    int qb_, bh_;
    if (!attn_block_map(p, qb_, bh_)) return;
    const int b = bh_ / p.H, h = bh_ % p.H;
    const int T = p.T;
    const long RS = 3L * p.H * 64, OS = (long)p.H * 64;
    const bf16_t* base = static_cast<const bf16_t*>(p.qkv) + (long)b * T * RS + h * 64;
    const bf16_t* gbase = static_cast<const bf16_t*>(p.dout) + (long)b * T * OS + h * 64;
    const int nthr = blockDim.x;
    const int q = qb_ * (nthr >> 1) + wave * 32 + ql;
    const bool qvalid = q < T;

    bf16x8 qr[4], gr[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        union { u32x4 v; bf16x8 hh; } uq, ug;
        uq.v = (u32x4){0u, 0u, 0u, 0u}; ug.v = uq.v;
        if (qvalid) {
            uq.v = *reinterpret_cast<const u32x4*>(base + (long)q * RS + (2 * kk + h2) * 8);
            ug.v = *reinterpret_cast<const u32x4*>(gbase + (long)q * OS + (2 * kk + h2) * 8);
        }
        qr[kk] = uq.hh; gr[kk] = ug.hh;
    }
    const float lse = qvalid ? p.lse[((long)b * p.H + h) * T + q] : 1e30f;
    const float del = qvalid ? p.delta[((long)b * p.H + h) * T + q] : 0.f;
    const float scale = p.scale_log2e * 0.6931471805599453f;
    f32x16 dq[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[i][r] = 0.f;
    const int a16 = lane & 15, g16 = (lane >> 4) & 1;

    const int HD = p.H * 64;
    kv_direct(base, RS, HD, 0, T, ldsK, ldsV, KP16, tid, nthr);
    kbias_fill(kbias, p.mask, b, 0, T, tid);
    __syncthreads();
    asm volatile("" ::"v"(qr[0]), "v"(qr[1]), "v"(qr[2]), "v"(qr[3]), "v"(gr[0]), "v"(gr[1]), "v"(gr[2]), "v"(gr[3]));   // see the forward kernel
    for (int kv0 = 0; kv0 < T; kv0 += KT) {
        const bool more = kv0 + KT < T;
        KVRegs nxt;
        if (more) kv_fetch(nxt, base, RS, HD, kv0 + KT, T, tid, nthr);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            if (kb == 1 && kv0 + 32 >= T) break;          // the second 32-key block is all padding
            // S^T[key][q], dP^T[key][q]: rows = keys (A from LDS), columns = queries (B = this lane's Q / dO row)
            bf16x8 ak[4], av[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                ak[kk] = ld_bf16x8(ldsK + (kb * 32 + ql) * KP16 + (2 * kk + h2) * 16);
                av[kk] = ld_bf16x8(ldsV + (kb * 32 + ql) * KP16 + (2 * kk + h2) * 16);
            }
            __builtin_amdgcn_sched_barrier(0);
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                s = SS_MFMA_32x32x16(ak[kk], qr[kk], s, 0, 0, 0);
                dp = SS_MFMA_32x32x16(av[kk], gr[kk], dp, 0, 0, 0);
            }
            float4 b4[4];                                 // key bias of this lane's 16 key rows
#pragma unroll
            for (int j = 0; j < 4; ++j) b4[j] = *reinterpret_cast<const float4*>(kbias + kb * 32 + 8 * j + 4 * h2);
            bf16x8 kf[4];                                 // K^T fragments [d][key-slots], index s2 * 2 + db
#pragma unroll
            for (int i = 0; i < 4; ++i)
                kf[i] = tr_frag(ldsK + (kb * 32 + 16 * (i >> 1) + 4 * h2 + (a16 >> 2)) * KP16 + ((i & 1) * 32 + 16 * g16 + 4 * (a16 & 3)) * 2,
                                8 * KP16);
            __builtin_amdgcn_sched_barrier(0);
            float ds[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float kbv = reinterpret_cast<const float*>(&b4[r >> 2])[r & 3];
                const float pr = __builtin_amdgcn_exp2f(fminf(s[r] * p.scale_log2e + kbv - lse, 0.f));
                float keep = 1.f;
                if (DROP) {
                    const int kk_ = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2;
                    const unsigned idx = ((unsigned)bh_ * T + q) * T + (kv0 + kk_);
                    keep = dropout_keep32(seed_fold(p.drop_seed), idx, p.drop_thresh) ? p.drop_scale : 0.f;
                }
                ds[r] = pr * (dp[r] * keep - del) * scale;
            }
            // dQ^T[d][q] += K^T[d][keys] . dS^T[keys][q]
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 df;
#pragma unroll
                for (int e = 0; e < 8; ++e) df[e] = (bf16_t)ds[8 * s2 + e];
#pragma unroll
                for (int db = 0; db < 2; ++db) dq[db] = SS_MFMA_32x32x16(kf[s2 * 2 + db], df, dq[db], 0, 0, 0);
            }
        }
        if (more) {
            __syncthreads();
            kv_commit(nxt, ldsK, ldsV, KP16, tid, nthr);
            kbias_fill(kbias, p.mask, b, kv0 + KT, T, tid);
            __syncthreads();
        }
    }
    // dq accumulators (transposed): column = this lane's query, rows d = db*32 + (r%4) + 8*(r/4) + 4*h2
    if (qvalid) {
        bf16_t* drow = static_cast<bf16_t*>(p.dqkv) + (long)b * T * RS + h * 64 + (long)q * RS;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
                store_bf16x4(drow + db * 32 + 8 * r4 + 4 * h2, dq[db][4 * r4], dq[db][4 * r4 + 1], dq[db][4 * r4 + 2], dq[db][4 * r4 + 3]);
    }
}

// ------------------------------------------------------------------------------------------------
// bf16 backward, short sequences (T <= 256): "resident" forms of the two passes.  The streaming kernels above move one
// 32-query (or 64-key) tile per iteration through LDS behind a register prefetch of ONE tile and a workgroup barrier: with ~600
// cycles of MFMA work per iteration and 2-5 k cycles of global latency under load, every iteration waits for its tile (17 us per
// block of 7 iterations at T = 197).  Here a block is one (batch, head); the operand that is walked - Q and dO for the dK/dV pass,
// K and V for the dQ pass - is copied whole into LDS by global_load_lds once (rows padded to a multiple of 32 by repeating row
// T-1: their lse = +inf / key bias = -inf makes every probability exactly 0), and each wave then runs over all tiles of it with no
// barrier, no staging and no register prefetch in the loop.
// One swizzle serves both ways these tiles are read (k-contiguous b128 fragments AND transposed ds_read_b64_tr_b16 fragments of
// the same image): 16-byte slot c of row r holds chunk c ^ ((3 * (r >> 1)) & 7): a bijection over eight consecutive row pairs
// (conflict-free b128 reads of 32 consecutive rows) whose neighbouring row pairs differ in more than bit 0 (the 4-row blocks of a
// transposed read do not collide either).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int qd_swz(int r) { return (3 * (r >> 1)) & 7; }

// copy `rows32` rows of 128 B (row stride `stride` elements, rows >= T repeat row T-1) into the swizzled LDS image at dst
__device__ __forceinline__ void res_copy_rows(const bf16_t* src, long stride, int T, int rows32, char* dst, int wave, int nw, int lane) {
    for (int piece = wave; piece < (rows32 >> 3); piece += nw) {
        const int r = piece * 8 + (lane >> 3);
        const int c = (lane & 7) ^ qd_swz(r);
        const int rr = r < T ? r : T - 1;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (long)rr * stride + c * 8),
                                         (__attribute__((address_space(3))) void*)(dst + piece * 1024), 16, 0, 2);      // nt: each row is read once
    }
}

__device__ __forceinline__ bf16x8 tr_frag2(const char* pa, const char* pb) {
    union { struct { s16x4 lo, hi; } h; bf16x8 v; } u;
    u.h.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(pa));
    u.h.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(pb));
    return u.v;
}

template <bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_res_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];      // [Q rows32 x 128][dO rows32 x 128][lse rows32][delta rows32]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h2 = lane >> 5, kl = lane & 31;
    const int bh_ = blockIdx.x;
    const int b = bh_ / p.H, h = bh_ % p.H;
    const int Tf = p.T, T = attn_teff(p, b);          // Tf: the tensors' row count; T: the rows this block works on
    if (p.trace && tid == 0) p.trace[(long)blockIdx.x * 4] = wall_clock64();
    const long RS = 3L * p.H * 64, OS = (long)p.H * 64;
    const bf16_t* base = static_cast<const bf16_t*>(p.qkv) + (long)b * Tf * RS + h * 64;
    const bf16_t* gbase = static_cast<const bf16_t*>(p.dout) + (long)b * Tf * OS + h * 64;
    const int nthr = blockDim.x, nw = nthr >> 6;
    const int q32 = (T + 31) / 32, rows32 = q32 * 32;
    char* ldsQ = lds;
    char* ldsG = lds + rows32 * 128;
    float* lse_l = reinterpret_cast<float*>(ldsG + rows32 * 128);
    float* del_l = lse_l + rows32;
    res_copy_rows(base, RS, T, rows32, ldsQ, wave, nw, lane);
    res_copy_rows(gbase, OS, T, rows32, ldsG, wave, nw, lane);
    for (int q = tid; q < rows32; q += nthr) {
        lse_l[q] = q < T ? p.lse[((long)b * p.H + h) * Tf + q] : 1e30f;            // -> P = 0 for padded queries
        del_l[q] = q < T ? p.delta[((long)b * p.H + h) * Tf + q] : 0.f;
    }
    const int a16 = lane & 15, g16 = (lane >> 4) & 1;
    const float scale = p.scale_log2e * 0.6931471805599453f;
    // fragment addressing inside a 32-query tile (tile start rows are multiples of 32: the swizzle terms depend on the lane only)
    const int arow = kl * 128, asw = qd_swz(kl);                                   // A operand rows = queries, 16-byte chunk 2 kk + h2
    const int trow = (4 * h2 + (a16 >> 2)) * 128, tsw = qd_swz(4 * h2 + (a16 >> 2));
    const int tch = g16 * 2 + ((a16 & 3) >> 1), tsub = ((a16 & 3) & 1) * 8;       // transposed: chunk db * 4 + tch, 8-byte half tsub
    bool landed = false;

#pragma unroll 1
    for (int kt = wave; kt < q32; kt += nw) {
        const int key = kt * 32 + kl;
        const bool kvalid = key < T;
        const float kb_ = (kvalid && (!p.mask || p.mask[(long)b * Tf + key] != 0)) ? 0.f : NEG;
        bf16x8 kr[4], vr[4];     // this lane's K and V row chunks: B operands of S = Q.K^T and dP = dO.V^T
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            union { u32x4 v; bf16x8 hh; } uk, uv;
            uk.v = (u32x4){0u, 0u, 0u, 0u}; uv.v = uk.v;
            if (kvalid) {
                uk.v = *reinterpret_cast<const u32x4*>(base + (long)key * RS + p.H * 64 + (2 * kk + h2) * 8);
                uv.v = *reinterpret_cast<const u32x4*>(base + (long)key * RS + 2 * p.H * 64 + (2 * kk + h2) * 8);
            }
            kr[kk] = uk.hh; vr[kk] = uv.hh;
        }
        if (!landed) {                  // first pass: the LDS image is complete once every wave's copies have landed
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            landed = true;
            if (p.trace && tid == 0) p.trace[(long)blockIdx.x * 4 + 1] = wall_clock64();
        }
        f32x16 dk[2], dv[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dk[i][r] = 0.f; dv[i][r] = 0.f; }
        for (int qt = 0; qt < q32; ++qt) {
            const int q0 = qt * 32;
            const char* cq = ldsQ + q0 * 128;
            const char* cg = ldsG + q0 * 128;
            bf16x8 aq[4], ag[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int off = arow + (((2 * kk + h2) ^ asw) << 4);
                aq[kk] = ld_bf16x8(cq + off);
                ag[kk] = ld_bf16x8(cg + off);
            }
            __builtin_amdgcn_sched_barrier(0);
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                s = SS_MFMA_32x32x16(aq[kk], kr[kk], s, 0, 0, 0);
                dp = SS_MFMA_32x32x16(ag[kk], vr[kk], dp, 0, 0, 0);
            }
            // requested while the MFMAs run: lse / delta of this lane's 16 query rows, and the transposed dO / Q fragments
            float4 l4[4], d4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                l4[j] = *reinterpret_cast<const float4*>(lse_l + q0 + 8 * j + 4 * h2);
                d4[j] = *reinterpret_cast<const float4*>(del_l + q0 + 8 * j + 4 * h2);
            }
            bf16x8 gf[4], qf[4];                              // index s2 * 2 + db
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ro = (16 * (i >> 1)) * 128 + trow;
                const int ca = ((((i & 1) * 4 + tch) ^ tsw) << 4) + tsub, cb = ((((i & 1) * 4 + tch) ^ tsw ^ 4) << 4) + tsub;
                gf[i] = tr_frag2(cg + ro + ca, cg + ro + 8 * 128 + cb);      // dO^T fragment: [d][q-slots]
                qf[i] = tr_frag2(cq + ro + ca, cq + ro + 8 * 128 + cb);      // Q^T fragment
            }
            __builtin_amdgcn_sched_barrier(0);
            // lane: key column kl, rows q = (r%4) + 8*(r/4) + 4*h2
            float pd[16], ds[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float lq = reinterpret_cast<const float*>(&l4[r >> 2])[r & 3];
                const float dq_ = reinterpret_cast<const float*>(&d4[r >> 2])[r & 3];
                const float pr = __builtin_amdgcn_exp2f(fminf(s[r] * p.scale_log2e + kb_ - lq, 0.f));
                float keep = 1.f;
                if (DROP) {
                    const int qq = (r & 3) + 8 * (r >> 2) + 4 * h2;
                    const unsigned idx = ((unsigned)bh_ * Tf + (q0 + qq)) * Tf + key;
                    keep = dropout_keep32(seed_fold(p.drop_seed), idx, p.drop_thresh) ? p.drop_scale : 0.f;
                }
                pd[r] = pr * keep;                                   // dropped probabilities feed dV
                ds[r] = pr * (dp[r] * keep - dq_) * scale;          // dS feeds dK
            }
            // dV^T[d][key] += dO^T[d][q] . P[q][key] ;  dK^T[d][key] += Q^T[d][q] . dS[q][key]
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 pf, df;
#pragma unroll
                for (int e = 0; e < 8; ++e) { pf[e] = (bf16_t)pd[8 * s2 + e]; df[e] = (bf16_t)ds[8 * s2 + e]; }
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    dv[db] = SS_MFMA_32x32x16(gf[s2 * 2 + db], pf, dv[db], 0, 0, 0);
                    dk[db] = SS_MFMA_32x32x16(qf[s2 * 2 + db], df, dk[db], 0, 0, 0);
                }
            }
        }
        // dk/dv accumulators (transposed): column = this lane's key, rows d = db*32 + (r%4) + 8*(r/4) + 4*h2
        if (kvalid) {
            bf16_t* drow = static_cast<bf16_t*>(p.dqkv) + (long)b * Tf * RS + h * 64 + (long)key * RS;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int d = db * 32 + 8 * r4 + 4 * h2;
                    store_bf16x4(drow + p.H * 64 + d, dk[db][4 * r4], dk[db][4 * r4 + 1], dk[db][4 * r4 + 2], dk[db][4 * r4 + 3]);
                    store_bf16x4(drow + 2 * p.H * 64 + d, dv[db][4 * r4], dv[db][4 * r4 + 1], dv[db][4 * r4 + 2], dv[db][4 * r4 + 3]);
                }
        }
    }
    if (!landed) {       // a wave without a tile (short effective length): its share of the copies still has to land before the others read
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (p.trace && tid == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); p.trace[(long)blockIdx.x * 4 + 2] = wall_clock64(); p.trace[(long)blockIdx.x * 4 + 3] = p.trace[(long)blockIdx.x * 4 + 2]; }
}

template <bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_res_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];      // [K rows32 x 128][V rows32 x 128][key bias rows32]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h2 = lane >> 5, ql = lane & 31;
    const int bh_ = blockIdx.x;
    const int b = bh_ / p.H, h = bh_ % p.H;
    const int Tf = p.T, T = attn_teff(p, b);          // Tf: the tensors' row count; T: the rows this block works on
    const long RS = 3L * p.H * 64, OS = (long)p.H * 64;
    const bf16_t* base = static_cast<const bf16_t*>(p.qkv) + (long)b * Tf * RS + h * 64;
    const bf16_t* gbase = static_cast<const bf16_t*>(p.dout) + (long)b * Tf * OS + h * 64;
    const int nthr = blockDim.x, nw = nthr >> 6;
    const int q32 = (T + 31) / 32, rows32 = q32 * 32;
    char* ldsK = lds;
    char* ldsV = lds + rows32 * 128;
    float* kbias = reinterpret_cast<float*>(ldsV + rows32 * 128);
    res_copy_rows(base + p.H * 64, RS, T, rows32, ldsK, wave, nw, lane);
    res_copy_rows(base + 2 * p.H * 64, RS, T, rows32, ldsV, wave, nw, lane);
    for (int key = tid; key < rows32; key += nthr)
        kbias[key] = (key < T && (!p.mask || p.mask[(long)b * Tf + key] != 0)) ? 0.f : NEG;
    const int a16 = lane & 15, g16 = (lane >> 4) & 1;
    const float scale = p.scale_log2e * 0.6931471805599453f;
    const int arow = ql * 128, asw = qd_swz(ql);
    const int trow = (4 * h2 + (a16 >> 2)) * 128, tsw = qd_swz(4 * h2 + (a16 >> 2));
    const int tch = g16 * 2 + ((a16 & 3) >> 1), tsub = ((a16 & 3) & 1) * 8;
    bool landed = false;

#pragma unroll 1
    for (int qt = wave; qt < q32; qt += nw) {
        const int q = qt * 32 + ql;
        const bool qvalid = q < T;
        bf16x8 qr[4], gr[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            union { u32x4 v; bf16x8 hh; } uq, ug;
            uq.v = (u32x4){0u, 0u, 0u, 0u}; ug.v = uq.v;
            if (qvalid) {
                uq.v = *reinterpret_cast<const u32x4*>(base + (long)q * RS + (2 * kk + h2) * 8);
                ug.v = *reinterpret_cast<const u32x4*>(gbase + (long)q * OS + (2 * kk + h2) * 8);
            }
            qr[kk] = uq.hh; gr[kk] = ug.hh;
        }
        const float lse = qvalid ? p.lse[((long)b * p.H + h) * Tf + q] : 1e30f;
        // delta = rowsum(dO o O) of this lane's query, from the dO slices it already holds (this kernel runs FIRST and leaves delta for
        // the dK / dV kernel: no separate pass over O and dO)
        float del = 0.f;
        if (qvalid) {
            const bf16_t* obase = static_cast<const bf16_t*>(p.out) + (long)b * Tf * OS + h * 64 + (long)q * OS;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const bf16x8 o8 = ld_bf16x8(obase + (2 * kk + h2) * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) del += (float)o8[e] * (float)gr[kk][e];
            }
        }
        del += __shfl_xor(del, 32, 64);
        if (qvalid && h2 == 0) const_cast<float*>(p.delta)[((long)b * p.H + h) * Tf + q] = del;
        if (!landed) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            landed = true;
        }
        f32x16 dq[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) dq[i][r] = 0.f;
        for (int kt = 0; kt < q32; ++kt) {
            const int k0 = kt * 32;
            const char* ck = ldsK + k0 * 128;
            const char* cv = ldsV + k0 * 128;
            // S^T[key][q], dP^T[key][q]: rows = keys (A from LDS), columns = queries (B = this lane's Q / dO row)
            bf16x8 ak[4], av[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int off = arow + (((2 * kk + h2) ^ asw) << 4);
                ak[kk] = ld_bf16x8(ck + off);
                av[kk] = ld_bf16x8(cv + off);
            }
            __builtin_amdgcn_sched_barrier(0);
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                s = SS_MFMA_32x32x16(ak[kk], qr[kk], s, 0, 0, 0);
                dp = SS_MFMA_32x32x16(av[kk], gr[kk], dp, 0, 0, 0);
            }
            float4 b4[4];                                 // key bias of this lane's 16 key rows
#pragma unroll
            for (int j = 0; j < 4; ++j) b4[j] = *reinterpret_cast<const float4*>(kbias + k0 + 8 * j + 4 * h2);
            bf16x8 kf[4];                                 // K^T fragments [d][key-slots], index s2 * 2 + db
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ro = (16 * (i >> 1)) * 128 + trow;
                const int ca = ((((i & 1) * 4 + tch) ^ tsw) << 4) + tsub, cb = ((((i & 1) * 4 + tch) ^ tsw ^ 4) << 4) + tsub;
                kf[i] = tr_frag2(ck + ro + ca, ck + ro + 8 * 128 + cb);
            }
            __builtin_amdgcn_sched_barrier(0);
            float ds[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float kbv = reinterpret_cast<const float*>(&b4[r >> 2])[r & 3];
                const float pr = __builtin_amdgcn_exp2f(fminf(s[r] * p.scale_log2e + kbv - lse, 0.f));
                float keep = 1.f;
                if (DROP) {
                    const int kk_ = k0 + (r & 3) + 8 * (r >> 2) + 4 * h2;
                    const unsigned idx = ((unsigned)bh_ * Tf + q) * Tf + kk_;
                    keep = dropout_keep32(seed_fold(p.drop_seed), idx, p.drop_thresh) ? p.drop_scale : 0.f;
                }
                ds[r] = pr * (dp[r] * keep - del) * scale;
            }
            // dQ^T[d][q] += K^T[d][keys] . dS^T[keys][q]
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 df;
#pragma unroll
                for (int e = 0; e < 8; ++e) df[e] = (bf16_t)ds[8 * s2 + e];
#pragma unroll
                for (int db = 0; db < 2; ++db) dq[db] = SS_MFMA_32x32x16(kf[s2 * 2 + db], df, dq[db], 0, 0, 0);
            }
        }
        if (qvalid) {
            bf16_t* drow = static_cast<bf16_t*>(p.dqkv) + (long)b * Tf * RS + h * 64 + (long)q * RS;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4)
                    store_bf16x4(drow + db * 32 + 8 * r4 + 4 * h2, dq[db][4 * r4], dq[db][4 * r4 + 1], dq[db][4 * r4 + 2], dq[db][4 * r4 + 3]);
        }
    }
    if (!landed) {       // a wave without a tile (short effective length): its share of the copies still has to land before the others read
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// bf16 backward, T <= 256 (ViT-B @224, every BERT caption): ONE kernel per (batch, head).  The two resident passes above each
// recompute S = Q.K^T, dP = dO.V^T and the exponentials, and each reads Q/K/V/dO from HBM (7 tile products and ~300 KB per head at
// T = 197).  Here wave w owns key tile w for dK / dV (K, V rows and the accumulators in registers, as in the dK/dV pass) AND query
// tile w for dQ.  It walks the query tiles in the order (w + j) mod n, j = 0..n-1, so at step j every wave holds the dS tile of a
// DIFFERENT query tile.  dS comes out of the MFMA with the key on the lane; dQ contracts over keys, so the bf16 dS tile takes the
// transposing trip of the GEMM epilogues anyway - staged [key][query], read back with ds_read_b64_tr_b16 as the B operand of
// dQ^T = K^T . dS^T - and that staging doubles as the hand-over: after the step's barrier wave w reads the tile that wave (w - j) mod n
// staged for query tile w, with the K^T fragments of that wave's key tile (transposed reads of the K image), and accumulates dQ in
// registers in a fixed order (deterministic).  Staging is double-buffered: one barrier per step.  5 tile products and ~200 KB per head.
// (A first form kept dQ as an fp32 image in LDS with b128 read-add-write per step: 16 KB of LDS traffic per tile, the 13-cycle wide
// stores made it LDS-bound - 675 us against 610 us for the two passes at B = 512, T = 197.)
// LDS: [Q rows32 x 128 B][dO][K][staging 2 x waves x 2304 B][lse][delta] = 137 KB at T = 256: one block of up to 8 waves per CU.
// ------------------------------------------------------------------------------------------------
constexpr int ONE_MAXT = 256;
constexpr int ONE_TP = 72;               // bytes per staged dS key row (32 queries x 2 B + 8: conflict-free 8-byte writes)
constexpr int ONE_ST = 32 * ONE_TP;      // one staged tile

__host__ __device__ inline int one_smem(int T) {
    const int q32 = (T + 31) / 32, rows32 = q32 * 32;
    return 3 * rows32 * 128 + 2 * q32 * ONE_ST + 2 * rows32 * 4 + 8 * 256;      // (+ the landing pad of the next head's touches, 256 B per wave)
}

template <bool DROP>
__global__ __launch_bounds__(512) void attn_bwd_one_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h2 = lane >> 5, kl = lane & 31;
    const int nthr = blockDim.x, nw = nthr >> 6;
    const int Tf = p.T;                               // the tensors' row count
    // Measured and dropped (tools/attn_bwd_ab.py, tools/dbg_attn_trace.py; B = 512, T = 197):
    //  - persistent blocks walking several heads, started a quarter period apart.  With one block per head the dispatcher keeps the CUs
    //    in step (all load, all compute, all store; a start offset on the first round is gone by the second); persistent blocks do keep
    //    their offsets, but a head's copies take ~5 us either way - 125 KB at the ~23 GB/s that one CU's outstanding requests sustain,
    //    not the HBM - so nothing is gained (481 vs 466 us).  (Their loop-invariant address arithmetic must also be kept from being
    //    hoisted: it spills, and scratch reloads put s_waitcnt vmcnt(0) between the requests - 686 us.)
    //  - an L2 prefetch of the next head through global_load_lds into a landing area nobody reads: the wait drops from 5.3 to 4.0 us,
    //    the tile loop grows from 10.0 to 13.3 us (LDS write traffic of the landing copies).
    const int bh_ = blockIdx.x;
    const int b = bh_ / p.H, h = bh_ % p.H;
    const int T = attn_teff(p, b);                    // the rows this block works on
    if (T <= 0) {                                     // an empty sequence of a ragged batch: no rows, zero column sums (uniform per block)
        if (p.colsum_ws)
            for (int i = tid; i < 192; i += nthr) p.colsum_ws[((long)b * 3 + i / 64) * (p.H * 64) + h * 64 + (i & 63)] = 0.f;
        return;
    }
    if (p.trace && tid == 0) p.trace[(long)bh_ * 4] = wall_clock64();
    const long RS = p.rs, OS = (long)p.H * 64, HP = p.H * p.ps;               // HP: from a head's q plane to its k plane, k to v
    const long row0 = p.row_start ? (long)p.row_start[b] : (long)b * Tf;      // first row of this sequence in qkv / out / dout / dqkv
    const bf16_t* base = static_cast<const bf16_t*>(p.qkv) + row0 * RS + h * p.ps;
    const bf16_t* gbase = static_cast<const bf16_t*>(p.dout) + row0 * OS + h * 64;
    const bf16_t* obase = static_cast<const bf16_t*>(p.out) + row0 * OS + h * 64;
    const int q32 = (T + 31) / 32, rows32 = q32 * 32;
    char* ldsQ = lds;
    char* ldsG = lds + rows32 * 128;
    char* ldsK = ldsG + rows32 * 128;
    char* stage = ldsK + rows32 * 128;
    float* lse_l = reinterpret_cast<float*>(stage + 2 * nw * ONE_ST);
    float* del_l = lse_l + rows32;
    const bool active = wave < q32;                   // (a block is launched with one wave per tile of the FULL length)
    res_copy_rows(base, RS, T, rows32, ldsQ, wave, nw, lane);
    res_copy_rows(gbase, OS, T, rows32, ldsG, wave, nw, lane);
    res_copy_rows(base + HP, RS, T, rows32, ldsK, wave, nw, lane);
    // every global request of the head goes out before anything waits: the value loads below are consumed after the copies' wait
    // (an LDS store of a loaded value right here would put a full s_waitcnt in front of the remaining requests: the resident passes
    // above pay two to three memory round trips that way, most of their 7.6 us 'waiting for copies')
    // (unconditional requests from clamped addresses, selected afterwards: a request inside a divergent branch also gets its own wait)
    const float lse_ld = p.lse[((long)b * p.H + h) * Tf + min(tid, T - 1)];
    // delta = rowsum(dO o O): eight lanes per query row, one 16-byte chunk each; O is requested here, dO is read from its LDS image
    const int per = nthr >> 3;                        // rows per pass; rows32 <= 4 * per (one wave per 32 rows)
    bf16x8 o8[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = (tid >> 3) + i * per;
        union { u32x4 v; bf16x8 hh; } uo;
        uo.v = *reinterpret_cast<const u32x4*>(obase + (long)min(q, T - 1) * OS + (tid & 7) * 8);
        o8[i] = uo.hh;
    }
    const int a16 = lane & 15, g16 = (lane >> 4) & 1, g4 = lane >> 4;
    const float scale = p.scale_log2e * 0.6931471805599453f;
    const int arow = kl * 128, asw = qd_swz(kl);
    const int trow = (4 * h2 + (a16 >> 2)) * 128, tsw = qd_swz(4 * h2 + (a16 >> 2));
    const int tch = g16 * 2 + ((a16 & 3) >> 1), tsub = ((a16 & 3) & 1) * 8;

    const int key = wave * 32 + kl;                   // this lane's key (dK / dV) and, for dQ, its query
    const bool kvalid = active && key < T;
    bf16x8 kr[4], vr[4];         // this lane's K and V row chunks: B operands of S = Q.K^T and dP = dO.V^T (K from its image, below)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        union { u32x4 v; bf16x8 hh; } uv;          // (keys >= T: row T-1, finite; their probabilities are exactly 0)
        uv.v = *reinterpret_cast<const u32x4*>(base + (long)min(key, T - 1) * RS + 2 * HP + (2 * kk + h2) * 8);
        vr[kk] = uv.hh;
    }
    long mk_ = 1;                // the last request: a wait the compiler attaches to this (uniform) branch coincides with the one below
    if (p.mask) mk_ = p.mask[(long)b * Tf + min(key, T - 1)];
    // dS staging: write [key = kl][4 queries 8 g + 4 h2 ..]; read back queries (g4 & 1) * 16 + a16 = lane % 32, key slots of MFMA step s2:
    // keys 16 s2 + 4 (lane / 32) + {0..3} (low half) and + 8 (high half) - the slot order of the transposed K fragments
    const int st_w = kl * ONE_TP + (4 * h2) * 2;
    const int st_r = (4 * (g4 >> 1) + (a16 >> 2)) * ONE_TP + ((g4 & 1) * 16 + (a16 & 3) * 4) * 2;
    typedef s16x4 __attribute__((address_space(3))) * lptr;

    f32x16 dk[2], dv[2], dq[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[i][r] = 0.f; dv[i][r] = 0.f; dq[i][r] = 0.f; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid < rows32) lse_l[tid] = tid < T ? -lse_ld : -1e30f;        // MINUS lse (one wave per 32 rows: rows32 <= blockDim.x); -1e30 -> P = 0 for padded queries
    const float kb_ = (kvalid && mk_ != 0) ? 0.f : NEG;
    const float kinit = kb_ / p.scale_log2e;          // S accumulators start here: fma(S, c, -lse) then carries the key bias
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = (tid >> 3) + i * per;
        float d = 0.f;
        if (q < T) {
            const bf16x8 g8 = ld_bf16x8(ldsG + q * 128 + (((tid & 7) ^ qd_swz(q)) << 4));
#pragma unroll
            for (int e = 0; e < 8; ++e) d += (float)o8[i][e] * (float)g8[e];
        }
        d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64);
        if ((tid & 7) == 0 && q < rows32) del_l[q] = -d;             // MINUS delta (rows >= T: 0)
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) kr[kk] = ld_bf16x8(ldsK + (active ? wave * 32 * 128 : 0) + arow + (((2 * kk + h2) ^ asw) << 4));
    // (the dropout instantiation re-reads these four fragments from the K image in every step instead of holding them: its hash
    // arithmetic needs the 16 registers - spilled, a V fragment's scratch round trip sits in the prologue and in every step)
    __syncthreads();
    if (p.trace && tid == 0) p.trace[(long)bh_ * 4 + 1] = wall_clock64();
    // With one block per CU the dispatcher keeps the CUs in step: all load (an HBM burst, matrix cores idle: 6.4 of a block's 17.5 us at
    // T = 197), all compute (HBM idle), all store.  The operands of THIS head are out of L2's way now (they sit in LDS / registers and are
    // read once), so the head that will take this CU's place - blockIdx.x + pf_stride, same XCD - has its operands touched here: one lane
    // per 128-byte row slice (a head's row of q, k, v, dO or O is exactly one cache line), 4 bytes each, landing in a 256-byte pad
    // nobody reads.  ~1000 lines = 16-20 wave-instructions per head, spread over the tile loop's 9 us instead of the next block's prologue;
    // issued from inline asm (the compiler's own wait counting never sees them: no vmcnt(0) in front of the step barriers).
    if (p.pf_stride > 0 && bh_ + p.pf_stride < p.B * p.H) {
        const int nh_ = bh_ + p.pf_stride;
        const int nb_ = nh_ / p.H, nhh_ = nh_ % p.H;
        const long nr_ = p.row_start ? (long)p.row_start[nb_] : (long)nb_ * Tf;
        const char* qb_ = reinterpret_cast<const char*>(p.qkv) + (nr_ * RS + nhh_ * p.ps) * 2;
        const char* gb_ = reinterpret_cast<const char*>(p.dout) + (nr_ * OS + nhh_ * 64) * 2;
        const char* ob_ = reinterpret_cast<const char*>(p.out) + (nr_ * OS + nhh_ * 64) * 2;
        // (rows32 comes out of shuffles / LDS in attn_teff: uniform, but not provably so - the M0 operand must be scalar)
        const unsigned pad_ = __builtin_amdgcn_readfirstlane(
            (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)(reinterpret_cast<char*>(del_l + rows32)) + wave * 256);
        const int groups_ = (Tf + 63) >> 6;                                       // wave-instructions per operand
        const int nparts_ = p.dbg == 5 ? 3 : (p.dbg == 6 ? 2 : (p.dbg == 7 ? 4 : 5));      // (A/B: how much of the head fits beside everything else in L2)
        for (int k = wave; k < nparts_ * groups_; k += nw) {
            const int part = k / groups_;
            const int r = min((k - part * groups_) * 64 + lane, Tf - 1);
            const char* sb_ = part < 3 ? qb_ + (long)part * (HP * 2) : (part == 3 ? gb_ : ob_);
            const unsigned voff = (unsigned)((long)r * (part < 3 ? RS : OS) * 2);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(sb_), "s"(pad_) : "memory");
        }
    }
#pragma unroll 1
    for (int j = 0; j < q32; ++j) {
        char* stj = stage + (j & 1) * nw * ONE_ST;
        if (active) {
            int qt = wave + j;
            if (qt >= q32) qt -= q32;
            const int q0 = qt * 32;
            const char* cq = ldsQ + q0 * 128;
            const char* cg = ldsG + q0 * 128;
            bf16x8 aq[4], ag[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int off = arow + (((2 * kk + h2) ^ asw) << 4);
                aq[kk] = ld_bf16x8(cq + off);
                ag[kk] = ld_bf16x8(cg + off);
                if (DROP) kr[kk] = ld_bf16x8(ldsK + wave * 32 * 128 + off);
            }
            __builtin_amdgcn_sched_barrier(0);
            // The accumulators start from the per-lane key bias (S) and from minus delta of their query rows (dP, straight from LDS), so the
            // element work below is x = fma(S, c, -lse), min, exp2, one multiply and the conversions: the exponential stage is VALU-bound
            // (v_exp_f32 issues at a quarter rate) and runs beside only one other wave's MFMAs.  The softmax scale of dS is applied to the
            // dK / dQ accumulators once, at the end.
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = kinit;
            if (DROP) {
#pragma unroll
                for (int r = 0; r < 16; ++r) dp[r] = 0.f;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 nd = *reinterpret_cast<const float4*>(del_l + q0 + 8 * i + 4 * h2);
                    dp[4 * i] = nd.x; dp[4 * i + 1] = nd.y; dp[4 * i + 2] = nd.z; dp[4 * i + 3] = nd.w;
                }
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                s = SS_MFMA_32x32x16(aq[kk], kr[kk], s, 0, 0, 0);
                dp = SS_MFMA_32x32x16(ag[kk], vr[kk], dp, 0, 0, 0);
            }
            // the rest in two halves of 16 query rows (MFMA step s2): lse / delta of the rows, the transposed dO / Q fragments, the
            // exponentials, then dV^T[d][key] += dO^T[d][q] . P[q][key] and dK^T[d][key] += Q^T[d][q] . dS[q][key].  (All 32 rows at
            // once held 32 more registers at the peak: spills, and a spilled V fragment puts scratch waits into the prologue.)
            char* mine = stj + wave * ONE_ST + st_w;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float4 l4[2], d4[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    l4[i] = *reinterpret_cast<const float4*>(lse_l + q0 + 8 * (2 * s2 + i) + 4 * h2);
                    if (DROP) d4[i] = *reinterpret_cast<const float4*>(del_l + q0 + 8 * (2 * s2 + i) + 4 * h2);
                }
                bf16x8 gf[2], qf[2];                          // index db
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const int ro = (16 * s2) * 128 + trow;
                    const int ca = (((db * 4 + tch) ^ tsw) << 4) + tsub, cb = (((db * 4 + tch) ^ tsw ^ 4) << 4) + tsub;
                    gf[db] = tr_frag2(cg + ro + ca, cg + ro + 8 * 128 + cb);      // dO^T fragment: [d][q-slots]
                    qf[db] = tr_frag2(cq + ro + ca, cq + ro + 8 * 128 + cb);      // Q^T fragment
                }
                __builtin_amdgcn_sched_barrier(0);
                // lane: key column kl, rows q = (r%4) + 8*(r/4) + 4*h2
                bf16x8 pf, df;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int r = 8 * s2 + e;
                    const float nlq = reinterpret_cast<const float*>(&l4[e >> 2])[e & 3];
                    const float pr = __builtin_amdgcn_exp2f(fminf(__builtin_fmaf(s[r], p.scale_log2e, nlq), 0.f));
                    float keep = 1.f;
                    if (DROP) {
                        const int qq = (r & 3) + 8 * (r >> 2) + 4 * h2;
                        const unsigned idx = ((unsigned)bh_ * Tf + (q0 + qq)) * Tf + key;
                        keep = dropout_keep32(seed_fold(p.drop_seed), idx, p.drop_thresh) ? p.drop_scale : 0.f;
                    }
                    pf[e] = (bf16_t)(pr * keep);                                   // dropped probabilities feed dV
                    if (DROP) df[e] = (bf16_t)(pr * (dp[r] * keep + reinterpret_cast<const float*>(&d4[e >> 2])[e & 3]));
                    else df[e] = (bf16_t)(pr * dp[r]);                             // dS / scale feeds dK and dQ
                }
#pragma unroll
                for (int g = 0; g < 2; ++g) {                        // the same bf16 dS, staged [key][query] for the wave that owns query tile qt
                    union { bf16_t hh[4]; uint2 u; } w;
#pragma unroll
                    for (int e = 0; e < 4; ++e) w.hh[e] = df[4 * g + e];
                    *reinterpret_cast<uint2*>(mine + (8 * (2 * s2 + g)) * 2) = w.u;
                }
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    dv[db] = SS_MFMA_32x32x16(gf[db], pf, dv[db], 0, 0, 0);
                    dk[db] = SS_MFMA_32x32x16(qf[db], df, dk[db], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
        if (active) {
            // dQ^T[d][q] += K^T[d][keys of tile kt] . dS^T[keys][q] for this wave's query tile: the tile wave kt staged in this step
            int kt = wave - j;
            if (kt < 0) kt += q32;
            const char* ck = ldsK + kt * 32 * 128;
            const char* src = stj + kt * ONE_ST + st_r;
            bf16x8 kf[4];                                     // index s2 * 2 + db
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ro = (16 * (i >> 1)) * 128 + trow;
                const int ca = ((((i & 1) * 4 + tch) ^ tsw) << 4) + tsub, cb = ((((i & 1) * 4 + tch) ^ tsw ^ 4) << 4) + tsub;
                kf[i] = tr_frag2(ck + ro + ca, ck + ro + 8 * 128 + cb);
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                union { struct { s16x4 lo, hi; } hq; bf16x8 v; } u;
                u.hq.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(src + s2 * 16 * ONE_TP));
                u.hq.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(src + s2 * 16 * ONE_TP + 8 * ONE_TP));
#pragma unroll
                for (int db = 0; db < 2; ++db) dq[db] = SS_MFMA_32x32x16(kf[s2 * 2 + db], u.v, dq[db], 0, 0, 0);
            }
        }
    }
    if (p.trace && tid == 0) p.trace[(long)bh_ * 4 + 2] = wall_clock64();
    // accumulators (transposed): column = this lane's key (dK, dV) / query (dQ), rows d = db*32 + (r%4) + 8*(r/4) + 4*h2.  Stored from
    // there, a wave-instruction writes 32 separate 16-byte segments (5 376 segments per head: 3.5 us of the per-block timeline).  Each
    // tile goes through the wave's share of the staging area instead - the lane writes its row's eight 8-byte pieces, the wave reads the
    // tile back 16 bytes per lane, row-contiguous - and leaves as whole 128-byte rows, eight per instruction.
    __syncthreads();             // every wave has left the last step: the staging area is free
    if (active) {
        constexpr int OP = 144;                       // bytes per staged row (128 + 16: 16-byte aligned rows)
        char* ob = stage + wave * (2 * ONE_ST);       // 32 x 144 = 4 608 B
        bf16_t* dbase = static_cast<bf16_t*>(p.dqkv) + row0 * RS + h * p.ps + (long)(wave * 32) * RS;
        float* cpart = reinterpret_cast<float*>(ldsQ);                 // [wave][3][64] column sums (the operand images are free by now)
        auto put = [&](const f32x16 (&acc)[2], int sel, float mul) {
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    union { bf16_t hh[4]; uint2 u; } w;
#pragma unroll
                    for (int e = 0; e < 4; ++e) w.hh[e] = (bf16_t)(acc[db][4 * r4 + e] * mul);
                    *reinterpret_cast<uint2*>(ob + kl * OP + (db * 32 + 8 * r4 + 4 * h2) * 2) = w.u;
                }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = (lane >> 3) + 8 * i;
                const u32x4 v = *reinterpret_cast<const u32x4*>(ob + row * OP + (lane & 7) * 16);
                if (wave * 32 + row < T) *reinterpret_cast<u32x4*>(dbase + (long)row * RS + sel * HP + (lane & 7) * 8) = v;
            }
            if (p.colsum_ws) {       // column sums of the tile's (rounded, as stored) valid rows: lane = 4 columns x every 4th row, 8-byte reads
                const int nrow = min(32, T - wave * 32), cg = lane & 15, rg = lane >> 4;
                float c[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int row = rg + 4 * i;
                    union { uint2 u; bf16_t hh[4]; } v;
                    v.u = *reinterpret_cast<const uint2*>(ob + row * OP + cg * 8);        // (unconditional: a branch per row serialises the eight reads)
                    const float keepf = row < nrow ? 1.f : 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) c[e] = __builtin_fmaf((float)v.hh[e], keepf, c[e]);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) { c[e] += __shfl_xor(c[e], 16, 64); c[e] += __shfl_xor(c[e], 32, 64); }
                if (rg == 0) *reinterpret_cast<float4*>(cpart + (wave * 3 + sel) * 64 + cg * 4) = make_float4(c[0], c[1], c[2], c[3]);
            }
            __builtin_amdgcn_wave_barrier();
        };
        put(dq, 0, scale);
        put(dk, 1, scale);
        put(dv, 2, 1.f);
    }
    if (p.colsum_ws) {           // this head's 3 x 64 column sums over its sequence: waves in a fixed order, plain stores (folded over B by the caller)
        // LDS-only barrier: __syncthreads() also waits for the block's global stores to be acknowledged (vmcnt(0)) - 3.7 us per block here,
        // which otherwise drain after the last wave has gone
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        for (int i = tid; i < 192; i += nthr) {
            const float* cpart = reinterpret_cast<const float*>(ldsQ);
            float c = 0.f;
            for (int w = 0; w < q32; ++w) c += cpart[(w * 3 + i / 64) * 64 + (i & 63)];
            p.colsum_ws[((long)b * 3 + i / 64) * (p.H * 64) + h * 64 + (i & 63)] = c;
        }
    }
    if (p.trace && tid == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); p.trace[(long)bh_ * 4 + 3] = wall_clock64(); }
}

template <bool DROP>
int launch_bwd_one(const AttnParams& p, hipStream_t stream) {
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_one_kernel<DROP>), hipFuncAttributeMaxDynamicSharedMemorySize, one_smem(ONE_MAXT));
        if (e != hipSuccess) return simseg_set_error("attention_bwd: cannot reserve LDS: %s", hipGetErrorString(e));
        configured = true;
    }
    const int q32 = (p.T + 31) / 32;
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = -1;
    }
    AttnParams q = p;
    // one block per CU is resident (LDS) for T > 128; shorter sequences fit two.  Variant 4 switches the next-head touches off (A/B runs).
    const int per_cu = one_smem(p.T) > 80 * 1024 ? 1 : 2;
    q.pf_stride = (cus > 0 && p.dbg != 4 && per_cu == 1 && (long)p.B * p.H > (long)cus) ? cus : 0;      // (two blocks per CU - T = 77 - measured 9 % slower with them)
    hipLaunchKernelGGL(attn_bwd_one_kernel<DROP>, dim3((unsigned)(p.B * p.H)), dim3(q32 * 64), one_smem(p.T), stream, q);
    return 0;
}

template <bool DROP>
int launch_bwd_res(const AttnParams& p, hipStream_t stream) {
    const int q32 = (p.T + 31) / 32, rows32 = q32 * 32;
    const int smem = 2 * rows32 * 128 + 2 * rows32 * 4;
    static bool configured = false;
    if (!configured) {
        const int mx = 2 * RES_MAXT * 128 + 2 * RES_MAXT * 4;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_res_kernel<DROP>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_res_kernel<DROP>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
        if (e != hipSuccess) return simseg_set_error("attention_bwd: cannot reserve LDS: %s", hipGetErrorString(e));
        configured = true;
    }
    const dim3 grid((unsigned)(p.B * p.H)), block((q32 < 4 ? q32 : 4) * 64);
    hipLaunchKernelGGL(attn_bwd_dq_res_kernel<DROP>, grid, block, smem, stream, p);       // dQ; computes and writes delta
    hipLaunchKernelGGL(attn_bwd_dkv_res_kernel<DROP>, grid, block, smem, stream, p);      // dK, dV (reads delta)
    return 0;
}

// Waves (= 32-query tiles) per block.  The kernels hold 2-3 waves per SIMD (register bound), i.e. 8-12 waves per CU, which
// 4-wave blocks tile exactly.  Measured on ViT-B (profiles/r1_kernel_roofline.txt): 4-wave blocks win at T = 197 (7 row tiles,
// one padded slot) AND at T = 1025 (33 row tiles: 0.109 ms against 0.136 ms for 3-wave blocks without padding and 0.164 ms
// for 6-wave blocks) - fewer waves per barrier and per staged K/V tile beat a perfectly filled last block.
int attn_waves_per_block(int q32) { return q32 <= 4 ? q32 : 4; }

thread_local unsigned long long* g_attn_trace = nullptr;
thread_local int g_attn_variant = 0;   // tests / benchmarks (thread-local selector): 1 = always the streaming (ring) kernels, 3 = backward as the two
                                       // resident passes, 4 = resident forward for every T <= 256; 102 / 103 = forward ablations (WRONG results:
                                       // no operand copies / no tile loop - tools/dbg_attn_res.py only)

int fill_params(AttnParams& p, const void* qkv, const int64_t* mask, int64_t B, int64_t T, int64_t H, float scale,
                uint64_t seed, float drop_p) {
    SS_CHECK(qkv, "attention: null qkv");
    SS_CHECK(B > 0 && T > 0 && H > 0 && B * H < 65536 * 16, "attention: bad shape");
    SS_CHECK(((uintptr_t)qkv % 16) == 0, "attention: qkv must be 16-byte aligned");
    SS_CHECK(drop_p >= 0.f && drop_p < 1.f, "attention: dropout p out of range");
    SS_CHECK(drop_p == 0.f || (double)B * H * T * T < 4294967296.0, "attention: dropout needs B*H*T*T < 2^32");
    memset(&p, 0, sizeof(p));
    p.qkv = qkv; p.mask = (const long*)mask; p.B = (int)B; p.T = (int)T; p.H = (int)H;
    p.scale_log2e = scale * 1.4426950408889634f;
    p.rs = 3L * H * 64; p.ps = 64;          // the packed projection rows; the *_planes entry points overwrite these
    p.dbg = g_attn_variant;
    p.trace = g_attn_trace;
    p.drop_seed = seed;
    p.drop_thresh = drop_p > 0.f ? (unsigned int)((double)drop_p * 4294967296.0) : 0u;
    p.drop_scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    return 0;
}

}  // namespace

// debug: resident blocks per CU the runtime predicts for the resident forward kernel at sequence length T
extern "C" int simseg_debug_attn_trace(void* buf) { g_attn_trace = static_cast<unsigned long long*>(buf); return 0; }

extern "C" int simseg_debug_attn_occupancy(int64_t T) {
    int n = -1;
    const int q32 = (int)((T + 31) / 32);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, attn_fwd_bf16_res_kernel<false, false>, (q32 < 4 ? q32 : 4) * 64, res_smem((int)T, false));
    return n;
}

extern "C" int simseg_set_attention_variant(int v) {
#ifndef SS_HALF
    simseg_set_attention_variant_h16(v);      // the fp16 flavour keeps its own (thread-local) selector
#endif
    g_w64_extra_lds = v == 8 ? 40000 : 0;     // 8 (tools/scratch/attn_w64_occ.py): the long-sequence forward at one block per CU
    g_attn_variant = v == 8 ? 0 : v;
    return 0;
}

// ctx[B,T,H*64] = softmax(q k^T * scale + keymask) v  from packed qkv[B,T,3,H,64]; lse[B,H,T] (log2 domain) optional.
extern "C" int simseg_attention_fwd(const void* qkv, const int64_t* key_mask, void* out, float* lse, int dtype, int64_t B,
                                    int64_t T, int64_t H, float scale, uint64_t drop_seed, float drop_p, int skip_padded_rows,
                                    void* stream) {
    SS_HALF_FWD(simseg_attention_fwd, qkv, key_mask, out, lse, dtype, B, T, H, scale, drop_seed, drop_p, skip_padded_rows, stream);
    AttnParams p;
    if (int rc = fill_params(p, qkv, key_mask, B, T, H, scale, drop_seed, drop_p)) return rc;
    p.pack = skip_padded_rows && key_mask && dtype == 1 && T <= RES_MAXT;
    SS_CHECK(out, "attention_fwd: null out");
    SS_CHECK(dtype == 0 || !(key_mask || drop_p > 0.f) || T <= MAXT_BIAS, "attention_fwd: masked bf16 sequences are limited to %d keys", MAXT_BIAS);
    p.out = out; p.lse = lse;
    // one wave per 32 queries; a block holds up to 8 waves of the same (batch, head) so K/V tiles are staged once
    const int q32 = (int)((T + 31) / 32);
    const int nw = attn_waves_per_block(q32);
    dim3 grid((unsigned)(((q32 + nw - 1) / nw) * (((B * H + 7) / 8) * 8)));      // 1-D, see attn_block_map
    if (dtype == 0)
        hipLaunchKernelGGL(attn_fwd_f32_kernel, grid, dim3(nw * 64), 0, (hipStream_t)stream, p);
    else if (T > 128 && T <= PERS_MAXT && !p.mask && !p.drop_thresh && !p.pack && (g_attn_variant == 4 || (g_attn_variant >= 40 && g_attn_variant <= 49))) {
        // the persistent loader-wave kernel: OPT-IN (variant 4; 41-43 = its timing ablations).  Measured slower than the one-block-per-head
        // resident kernel at the ViT-B training shape (223 vs 177 us, tools/attn_fwd_ab.py) - see the notes at the kernel
        const int rc = launch_fwd_pers(p, (hipStream_t)stream);
        if (rc) return rc;
    } else if ((g_attn_variant == 6 || g_attn_variant == 7) && T >= 65 && !p.mask && !p.drop_thresh) {
        launch_w64<false>(p, (hipStream_t)stream, g_attn_variant == 6 ? 1 : 2);      // tools: the long-sequence forward on a short sequence
    } else if ((T <= RES_AUTO_T || (g_attn_variant >= 2 && T <= RES_MAXT)) && g_attn_variant != 1) {
        int rc;
        if (p.drop_thresh) rc = launch_fwd_res<true, true>(p, (hipStream_t)stream);
        else if (p.mask) rc = launch_fwd_res<false, true>(p, (hipStream_t)stream);
        else rc = launch_fwd_res<false, false>(p, (hipStream_t)stream);
        if (rc) return rc;
    } else if (T >= W64_MINT && !p.mask && !p.drop_thresh && g_attn_variant != 1) {
        // 64 queries per wave, four waves per block, every block of a (batch, head) on one XCD
        launch_w64<false>(p, (hipStream_t)stream, g_attn_variant == 6 ? 1 : (g_attn_variant == 7 ? 2 : 0));
    } else
        if (p.drop_thresh) hipLaunchKernelGGL((attn_fwd_bf16_kernel<true, true, false>), grid, dim3(nw * 64), 0, (hipStream_t)stream, p);
        else if (p.mask) hipLaunchKernelGGL((attn_fwd_bf16_kernel<false, true, false>), grid, dim3(nw * 64), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((attn_fwd_bf16_kernel<false, false, false>), grid, dim3(nw * 64), 0, (hipStream_t)stream, p);
    SS_LAUNCH_CHECK("attention_fwd");
    return 0;
}

// The long-sequence 16-bit forward on operands whose q columns ALREADY carry scale * log2(e) (the caller folded the factor into the q rows
// of the projection weight and bias before rounding them to 16 bits: the product is rounded once, as q itself would have been, where the
// kernel's own scaling of a 16-bit q rounds a second time).  T >= 512, no mask, no dropout; lse (optional) in the log2 domain as above.
extern "C" int simseg_attention_fwd_qscaled(const void* qkv, void* out, float* lse, int64_t B, int64_t T, int64_t H, void* stream) {
    SS_HALF_FWD(simseg_attention_fwd_qscaled, qkv, out, lse, B, T, H, stream);
    AttnParams p;
    if (int rc = fill_params(p, qkv, nullptr, B, T, H, 1.0f, 0, 0.f)) return rc;
    SS_CHECK(out, "attention_fwd_qscaled: null out");
    SS_CHECK(T >= W64_MINT, "attention_fwd_qscaled: sequences of at least %d tokens", W64_MINT);
    p.out = out; p.lse = lse; p.qscaled = 1;
    launch_w64<false>(p, (hipStream_t)stream, g_attn_variant == 6 ? 1 : (g_attn_variant == 7 ? 2 : 0));
    SS_LAUNCH_CHECK("attention_fwd_qscaled");
    return 0;
}

// Exact-mode attention forward through the bf16 pieces of the fp32 operands (attn_fwd_x3_kernel).  qkv3: bf16 [3 pieces][B, T, 3, H, 64]
// (simseg_split_bf16x3 with b_pattern = 2 of the fp32 packed projection), piece p at qkv3 + p * plane_elems; out fp32 [B, T, H * 64].
extern "C" int simseg_attention_fwd_x3(const void* qkv3, int64_t plane_elems, float* out, int64_t B, int64_t T, int64_t H, float scale, void* stream) {
    SS_CHECK(qkv3 && out, "attention_fwd_x3: null pointer");
    SS_CHECK(B > 0 && T > 0 && H > 0 && B * H < 65536 * 16 && plane_elems >= B * T * 3 * H * 64, "attention_fwd_x3: bad shape");
    SS_CHECK(((uintptr_t)qkv3 % 16) == 0 && ((uintptr_t)out % 16) == 0 && plane_elems % 8 == 0, "attention_fwd_x3: operands must be 16-byte aligned");
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, X3_NS * X3_STAGE);
        if (e != hipSuccess) return simseg_set_error("attention_fwd_x3: cannot reserve LDS: %s", hipGetErrorString(e));
        configured = true;
    }
    AttnX3Params p;
    p.qkv3 = static_cast<const bf16_t*>(qkv3); p.plane = (long)plane_elems; p.out = out; p.B = (int)B; p.T = (int)T; p.H = (int)H;
    p.scale_log2e = scale * 1.4426950408889634f;
    p.dbg = g_attn_variant >= 200 ? g_attn_variant - 200 : 0;      // (timing ablations, WRONG results: tools/attn_x3_bench.py)
    // waves (32-query tiles) per block: 4..8, the count that leaves the fewest padding waves in a head's last block (33 tiles at
    // T = 1025: five blocks of 7; 11 at T = 325: two of 6); ties go to the larger block
    const int q32 = (int)((T + 31) / 32);
    int nw = q32 < 4 ? (q32 < 1 ? 1 : q32) : 4, best = 1 << 30;
    for (int c = 4; c <= 8 && q32 >= 4; ++c) {
        const int padded = ((q32 + c - 1) / c) * c;
        if (padded <= best) { best = padded; nw = c; }
    }
    const int gx = (q32 + nw - 1) / nw;
    hipLaunchKernelGGL(attn_fwd_x3_kernel, dim3((unsigned)(gx * (((B * H + 7) / 8) * 8))), dim3(nw * 64), X3_NS * X3_STAGE, (hipStream_t)stream, p);
    SS_LAUNCH_CHECK("attention_fwd_x3");
    return 0;
}


// debug: forward with a cycle-counter timeline of block (0,0) / thread 0 written to dbg[0..4] (5 x u64):
// {first tile staging, S = K.Q^T, softmax + rescale, P.V, next-tile commit + barriers}, summed over the K/V tiles
extern "C" int simseg_debug_attention_timeline(const void* qkv, void* out, float* lse, void* dbg, int64_t B, int64_t T, int64_t H, void* stream) {
    AttnParams p;
    if (int rc = fill_params(p, qkv, nullptr, B, T, H, 0.125f, 0, 0.f)) return rc;
    p.out = out; p.lse = lse; p.delta = reinterpret_cast<const float*>(dbg);
    const int q32 = (int)((T + 31) / 32);
    const int nw = attn_waves_per_block(q32);
    dim3 grid((unsigned)(((q32 + nw - 1) / nw) * (((B * H + 7) / 8) * 8)));      // 1-D, see attn_block_map
    if (T >= W64_MINT && (T - 1) % 64 == 0 && g_attn_variant != 1)      // the w64 kernel's record layout: see the end of attn_fwd_w64_kernel
        { p.qscaled = 1; launch_w64<true>(p, (hipStream_t)stream, g_attn_variant == 6 ? 1 : (g_attn_variant == 7 ? 2 : 0)); }      // (timing only: q taken as pre-scaled)
    else
        hipLaunchKernelGGL((attn_fwd_bf16_kernel<false, false, true>), grid, dim3(nw * 64), 0, (hipStream_t)stream, p);
    SS_LAUNCH_CHECK("attention_timeline");
    return 0;
}

// backward: dqkv[B,T,3,H,64] from qkv, ctx (forward output), dctx, lse; delta[B,H,T] is caller-provided scratch.  All tensors in
// `dtype` (0 = fp32 exact mode, 1 = bf16).
extern "C" int64_t simseg_attention_bwd_workspace_bytes(int64_t B, int64_t T, int64_t H) {
    return (B * H * T + B * 3 * H * 64) * (int64_t)sizeof(float);      // delta[B,H,T] + per-sequence column sums [B, 3*H*64]
}

extern "C" int simseg_colsum_accum(const void* in, int in_dtype, float* out, int64_t rows, int64_t N, int64_t ld, void* stream);

extern "C" int simseg_attention_bwd(const void* qkv, const int64_t* key_mask, const void* out, const void* dout, const float* lse,
                                    float* workspace, void* dqkv, float* dqkv_colsum, int dtype, int64_t B, int64_t T, int64_t H, float scale,
                                    uint64_t drop_seed, float drop_p, int skip_padded_rows, void* stream) {
    SS_HALF_FWD(simseg_attention_bwd, qkv, key_mask, out, dout, lse, workspace, dqkv, dqkv_colsum, dtype, B, T, H, scale, drop_seed, drop_p, skip_padded_rows, stream);
    float* delta = workspace;
    AttnParams p;
    if (int rc = fill_params(p, qkv, key_mask, B, T, H, scale, drop_seed, drop_p)) return rc;
    p.pack = skip_padded_rows && key_mask && dtype == 1 && T <= RES_MAXT;
    SS_CHECK(out && dout && lse && workspace && dqkv, "attention_bwd: null pointer");
    SS_CHECK(dtype == 0 || dtype == 1, "attention_bwd: dtype must be 0 (fp32) or 1 (bf16)");
    p.out = const_cast<void*>(out); p.dout = dout; p.lse = const_cast<float*>(lse); p.delta = delta; p.dqkv = dqkv;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == 0) {
        SS_CHECK(((uintptr_t)out % 16) == 0 && ((uintptr_t)dout % 16) == 0 && ((uintptr_t)dqkv % 16) == 0, "attention_bwd: operands must be 16-byte aligned");
        const int q32 = (int)((T + 31) / 32);
        const int nw = attn_waves_per_block(q32);
        dim3 grid((unsigned)(((q32 + nw - 1) / nw) * (((B * H + 7) / 8) * 8)));      // 1-D, see attn_block_map
        if (p.drop_thresh) {
            hipLaunchKernelGGL((attn_bwd_f32_kernel<false, true>), grid, dim3(nw * 64), 0, s, p);       // dQ, writes delta
            hipLaunchKernelGGL((attn_bwd_f32_kernel<true, true>), grid, dim3(nw * 64), 0, s, p);        // dK, dV
        } else {
            hipLaunchKernelGGL((attn_bwd_f32_kernel<false, false>), grid, dim3(nw * 64), 0, s, p);
            hipLaunchKernelGGL((attn_bwd_f32_kernel<true, false>), grid, dim3(nw * 64), 0, s, p);
        }
        SS_LAUNCH_CHECK("attention_bwd (fp32)");
        if (dqkv_colsum) return simseg_colsum_accum(dqkv, 0, dqkv_colsum, B * T, 3 * H * 64, 3 * H * 64, stream);
        return 0;
    }
    const long groups = B * T * H;
    const int q32 = (int)((T + 31) / 32);
    const int nw = attn_waves_per_block(q32);
    dim3 grid((unsigned)(((q32 + nw - 1) / nw) * (((B * H + 7) / 8) * 8)));      // 1-D, see attn_block_map
    const bool resident = T <= RES_MAXT && g_attn_variant != 1;
    if (!resident)       // (the resident dQ kernel forms delta itself)
        hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((groups * 8 + 255) / 256)), dim3(256), 0, s, (const bf16_t*)out,
                           (const bf16_t*)dout, delta, (int)B, (int)T, (int)H);
    if (resident && T <= ONE_MAXT && g_attn_variant != 3) {       // (variant 3: the two resident passes, for A/B runs)
        // the qkv bias gradient rides along: per-sequence column sums from the kernel, folded over the batch by the column-sum kernel
        // ([B, 3*H*64] fp32: 4.7 MB at the training shape instead of a pass over the 465 MB of dqkv)
        p.colsum_ws = dqkv_colsum ? workspace + B * H * T : nullptr;
        if (int rc = p.drop_thresh ? launch_bwd_one<true>(p, s) : launch_bwd_one<false>(p, s)) return rc;
        SS_LAUNCH_CHECK("attention_bwd");
        if (dqkv_colsum) return simseg_colsum_accum(p.colsum_ws, 0, dqkv_colsum, B, 3 * H * 64, 3 * H * 64, stream);
        return 0;
    }
    if (dqkv_colsum && p.pack) hipMemsetAsync(dqkv, 0, (size_t)B * T * 3 * H * 64 * 2, s);     // (rows these kernels leave untouched are summed below)
    if (resident) {
        if (int rc = p.drop_thresh ? launch_bwd_res<true>(p, s) : launch_bwd_res<false>(p, s)) return rc;
    } else if (p.drop_thresh) {
        hipLaunchKernelGGL(attn_bwd_dkv_kernel<true>, grid, dim3(nw * 64), 0, s, p);
        hipLaunchKernelGGL(attn_bwd_dq_kernel<true>, grid, dim3(nw * 64), 0, s, p);
    } else {
        hipLaunchKernelGGL(attn_bwd_dkv_kernel<false>, grid, dim3(nw * 64), 0, s, p);
        hipLaunchKernelGGL(attn_bwd_dq_kernel<false>, grid, dim3(nw * 64), 0, s, p);
    }
    SS_LAUNCH_CHECK("attention_bwd");
    if (dqkv_colsum) return simseg_colsum_accum(dqkv, 1, dqkv_colsum, B * T, 3 * H * 64, 3 * H * 64, stream);
    return 0;
}

// ---- ragged batches stored without their padding (the packed text tower) -------------------------------------------------------
// qkv [rows, 3, H, 64], out / dout [rows, H*64], dqkv like qkv: sequence b is rows [row_start[b], row_start[b+1]) (row_start: B + 1
// int32 on the device, ascending), at most T tokens (T <= 256), every stored token real - attention has no notion of position, so
// this is the masked dense computation on the unmasked tokens (HF BertSelfAttention with a key-padding mask; the reference's captions
// are prefix-masked).  lse [B, H, T].  The dropout hash is indexed by (sequence, head, query index, key index) with stride T as in the
// dense entry points: for prefix masks the same probabilities are dropped.  bf16 only.
extern "C" int simseg_attention_fwd_rows(const void* qkv, const int32_t* row_start, void* out, float* lse, int64_t B, int64_t T, int64_t H,
                                         float scale, uint64_t drop_seed, float drop_p, void* stream) {
    SS_HALF_FWD(simseg_attention_fwd_rows, qkv, row_start, out, lse, B, T, H, scale, drop_seed, drop_p, stream);
    AttnParams p;
    if (int rc = fill_params(p, qkv, nullptr, B, T, H, scale, drop_seed, drop_p)) return rc;
    SS_CHECK(out && row_start, "attention_fwd_rows: null pointer");
    SS_CHECK(T <= RES_MAXT, "attention_fwd_rows: sequences of at most %d tokens (got %lld)", RES_MAXT, (long long)T);
    p.out = out; p.lse = lse; p.row_start = row_start;
    int rc = p.drop_thresh ? launch_fwd_res<true, true>(p, (hipStream_t)stream) : launch_fwd_res<false, false>(p, (hipStream_t)stream);
    if (rc) return rc;
    SS_LAUNCH_CHECK("attention_fwd_rows");
    return 0;
}

extern "C" int simseg_attention_bwd_rows(const void* qkv, const int32_t* row_start, const void* out, const void* dout, const float* lse,
                                         float* workspace, void* dqkv, float* dqkv_colsum, int64_t B, int64_t T, int64_t H, float scale,
                                         uint64_t drop_seed, float drop_p, void* stream) {
    SS_HALF_FWD(simseg_attention_bwd_rows, qkv, row_start, out, dout, lse, workspace, dqkv, dqkv_colsum, B, T, H, scale, drop_seed, drop_p, stream);
    AttnParams p;
    if (int rc = fill_params(p, qkv, nullptr, B, T, H, scale, drop_seed, drop_p)) return rc;
    SS_CHECK(out && dout && lse && workspace && dqkv && row_start, "attention_bwd_rows: null pointer");
    SS_CHECK(T <= ONE_MAXT, "attention_bwd_rows: sequences of at most %d tokens (got %lld)", ONE_MAXT, (long long)T);
    p.out = const_cast<void*>(out); p.dout = dout; p.lse = const_cast<float*>(lse); p.delta = workspace; p.dqkv = dqkv; p.row_start = row_start;
    hipStream_t s = (hipStream_t)stream;
    p.colsum_ws = dqkv_colsum ? workspace + B * H * T : nullptr;
    if (int rc = p.drop_thresh ? launch_bwd_one<true>(p, s) : launch_bwd_one<false>(p, s)) return rc;
    SS_LAUNCH_CHECK("attention_bwd_rows");
    if (dqkv_colsum) return simseg_colsum_accum(p.colsum_ws, 0, dqkv_colsum, B, 3 * H * 64, 3 * H * 64, stream);
    return 0;
}

// ---- plane-major projection operands (round 4) -----------------------------------------------------------------------------------
// qkv / dqkv as [3 * H][plane_rows][64]: plane which * H + h holds the 64 channels of head h of q (which = 0), k (1) or v (2) for every
// token row - what the qkv GEMM's plane-wise epilogue writes and what the dgrad / wgrad GEMMs read with a K-tile stride.  A head's operand
// rows are then ONE contiguous run (25 KB at T = 197) instead of T lines 4.6 KB apart.  Sequence b = rows [b * T, (b + 1) * T) when
// row_start is null (dense batch, no key mask), else rows [row_start[b], row_start[b + 1]) as in the *_rows entry points.  T <= 256, 16-bit only.
extern "C" int simseg_attention_fwd_planes(const void* qkv, int64_t plane_rows, const int32_t* row_start, void* out, float* lse, int64_t B, int64_t T,
                                           int64_t H, float scale, uint64_t drop_seed, float drop_p, void* stream) {
    SS_HALF_FWD(simseg_attention_fwd_planes, qkv, plane_rows, row_start, out, lse, B, T, H, scale, drop_seed, drop_p, stream);
    AttnParams p;
    if (int rc = fill_params(p, qkv, nullptr, B, T, H, scale, drop_seed, drop_p)) return rc;
    SS_CHECK(out, "attention_fwd_planes: null pointer");
    SS_CHECK(T <= RES_MAXT, "attention_fwd_planes: sequences of at most %d tokens (got %lld)", RES_MAXT, (long long)T);
    SS_CHECK(plane_rows >= (row_start ? 1 : B * T), "attention_fwd_planes: plane_rows %lld is smaller than the batch", (long long)plane_rows);
    p.out = out; p.lse = lse; p.row_start = row_start; p.rs = 64; p.ps = plane_rows * 64;
    int rc = p.drop_thresh ? launch_fwd_res<true, true>(p, (hipStream_t)stream) : launch_fwd_res<false, false>(p, (hipStream_t)stream);
    if (rc) return rc;
    SS_LAUNCH_CHECK("attention_fwd_planes");
    return 0;
}

extern "C" int simseg_attention_bwd_planes(const void* qkv, int64_t plane_rows, const int32_t* row_start, const void* out, const void* dout,
                                           const float* lse, float* workspace, void* dqkv, float* dqkv_colsum, int64_t B, int64_t T, int64_t H,
                                           float scale, uint64_t drop_seed, float drop_p, void* stream) {
    SS_HALF_FWD(simseg_attention_bwd_planes, qkv, plane_rows, row_start, out, dout, lse, workspace, dqkv, dqkv_colsum, B, T, H, scale, drop_seed, drop_p, stream);
    AttnParams p;
    if (int rc = fill_params(p, qkv, nullptr, B, T, H, scale, drop_seed, drop_p)) return rc;
    SS_CHECK(out && dout && lse && workspace && dqkv, "attention_bwd_planes: null pointer");
    SS_CHECK(T <= ONE_MAXT, "attention_bwd_planes: sequences of at most %d tokens (got %lld)", ONE_MAXT, (long long)T);
    SS_CHECK(plane_rows >= (row_start ? 1 : B * T), "attention_bwd_planes: plane_rows %lld is smaller than the batch", (long long)plane_rows);
    p.out = const_cast<void*>(out); p.dout = dout; p.lse = const_cast<float*>(lse); p.delta = workspace; p.dqkv = dqkv; p.row_start = row_start;
    p.rs = 64; p.ps = plane_rows * 64;
    hipStream_t s = (hipStream_t)stream;
    p.colsum_ws = dqkv_colsum ? workspace + B * H * T : nullptr;
    if (int rc = p.drop_thresh ? launch_bwd_one<true>(p, s) : launch_bwd_one<false>(p, s)) return rc;
    SS_LAUNCH_CHECK("attention_bwd_planes");
    if (dqkv_colsum) return simseg_colsum_accum(p.colsum_ws, 0, dqkv_colsum, B, 3 * H * 64, 3 * H * 64, stream);
    return 0;
}
