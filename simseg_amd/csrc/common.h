// Shared device/host helpers for libsimseg_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

// The 16-bit operand type.  gemm.hip, attn.hip, rowops.hip and loss.hip are compiled TWICE (simseg_amd/build.py): as they stand - bf16, the
// headline mode - and with -DSS_HALF, where the same sources are the IEEE fp16 flavour the reference's AMP mode uses (fp16 autocast + a
// live GradScaler, clip_runner.py:226-230): `bf16_t` is then _Float16, the MFMA is v_mfma_f32_32x32x16_f16, and every C-ABI entry point
// the file defines carries the suffix _h16 (half_names.h renames them by macro).  The bf16 build's entry points hand a call over to their
// _h16 twin while the calling thread has selected fp16 (simseg_set_half_type(2)); dtype code 1 always means "the selected 16-bit type".
#ifdef SS_HALF
typedef _Float16 bf16_t;
#define SS_MFMA_32x32x16(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z)
#else
typedef __bf16 bf16_t;
#define SS_MFMA_32x32x16(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z)
#endif
typedef __attribute__((ext_vector_type(8))) bf16_t bf16x8;
typedef __attribute__((ext_vector_type(4))) bf16_t bf16x4;
#include "half_names.h"
#ifndef SS_HALF
extern thread_local int g_ss_half;          // 1 = bf16 (default), 2 = fp16: simseg_set_half_type
#define SS_HALF_FWD(name, ...)                                  \
    do {                                                        \
        if (g_ss_half == 2) return name##_h16(__VA_ARGS__);     \
    } while (0)
#else
#define SS_HALF_FWD(name, ...)
#endif
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define WAVE 64

// ---- error plumbing (no exceptions cross the C ABI) -------------------------------------------
extern thread_local char g_simseg_err[512];
int simseg_set_error(const char* fmt, ...);

#define SS_CHECK(cond, ...)                          \
    do {                                             \
        if (!(cond)) return simseg_set_error(__VA_ARGS__); \
    } while (0)

#define SS_LAUNCH_CHECK(name)                                                         \
    do {                                                                              \
        hipError_t e__ = hipGetLastError();                                           \
        if (e__ != hipSuccess) return simseg_set_error("%s: %s", name, hipGetErrorString(e__)); \
    } while (0)

// ---- numeric helpers --------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t v) { return (float)v; }
__device__ __forceinline__ float half_bits_to_float(short bits) {      // a 16-bit operand that travelled as raw bits (transposing LDS reads)
    union { short s; bf16_t h; } u;
    u.s = bits;
    return (float)u.h;
}
__device__ __forceinline__ bf16_t f2bf(float v) { return (bf16_t)v; }

// Normal CDF for the erf-GELU epilogues.  libm's erff costs ~30 VALU instructions per element, which made the GELU / GELU'
// epilogues of the MLP GEMMs as long as their K loops (measured: +40 % / +60 % on [100864,768]x[3072,768]^T).  Abramowitz &
// Stegun 7.1.26 needs one exp, one reciprocal and five FMAs, and its exp(-x^2/2) is the Gaussian the derivative needs anyway:
//   erf(z) = 1 - (a1 t + ... + a5 t^5) e^{-z^2},  t = 1 / (1 + p z),  z >= 0,   |error| <= 1.5e-7
// The lower tail is formed as 0.5 * poly * e directly (no 1 - 1 cancellation).  `gauss` returns exp(-x^2/2).
__device__ __forceinline__ float norm_cdf(float x, float& gauss) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
    gauss = __expf(-z * z);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float tail = 0.5f * poly * gauss;               // = 0.5 * erfc(|x| / sqrt 2)
    return x >= 0.f ? 1.0f - tail : tail;
}
__device__ __forceinline__ float gelu_erf(float x) {
    float g;
    return x * norm_cdf(x, g);
}
// y = gelu(x) and dy = gelu'(x) from one CDF / Gaussian evaluation (forward epilogue that stores the derivative for backward)
__device__ __forceinline__ float gelu_erf_grad(float x, float& dy) {
    float g;
    const float cdf = norm_cdf(x, g);
    dy = cdf + x * 0.39894228040143268f * g;
    return x * cdf;
}
// bf16 epilogues only: erf-GELU and its derivative from an odd degree-15 polynomial fit of the normal CDF on |x| <= 3.5 (argument
// clamped beyond: the CDF is then within 2.4e-4 of 0 / 1), no transcendental instruction.  v_exp / v_rcp issue at a quarter of the FMA
// rate, and the fc1-forward epilogue (GELU + GELU' of 64 K values per tile) was bound by exactly that arithmetic: 66 -> ~40 issue cycles
// per element.  Fit error (fp32 Horner): CDF 7e-6, GELU 1.3e-5, GELU' 4.4e-4 - below bf16's 4e-3 relative rounding of the stored values.
// The exact-fp32 path keeps norm_cdf above.
__device__ __forceinline__ float gelu_poly_grad(float x, float& dy) {
    const float xc = __builtin_amdgcn_fmed3f(x, -3.5f, 3.5f);
    const float s = xc * xc;
    float q = -2.815794181e-09f, dq = -4.223691272e-08f;
    q = fmaf(q, s, 1.818798183e-07f);   dq = fmaf(dq, s, 2.364437638e-06f);
    q = fmaf(q, s, -5.270657400e-06f);  dq = fmaf(dq, s, -5.797723140e-05f);
    q = fmaf(q, s, 9.220430572e-05f);   dq = fmaf(dq, s, 8.298387515e-04f);
    q = fmaf(q, s, -1.108776002e-03f);  dq = fmaf(dq, s, -7.761432013e-03f);
    q = fmaf(q, s, 9.826695057e-03f);   dq = fmaf(dq, s, 4.913347528e-02f);
    q = fmaf(q, s, -6.636358108e-02f);  dq = fmaf(dq, s, -1.990907432e-01f);
    q = fmaf(q, s, 3.989096663e-01f);   dq = fmaf(dq, s, 3.989096663e-01f);
    const float cdf = fmaf(xc, q, 0.5f);
    dy = fmaf(xc, dq, cdf);
    return x * cdf;
}
__device__ __forceinline__ float gelu_poly(float x) {
    const float xc = __builtin_amdgcn_fmed3f(x, -3.5f, 3.5f);
    const float s = xc * xc;
    float q = -2.815794181e-09f;
    q = fmaf(q, s, 1.818798183e-07f);
    q = fmaf(q, s, -5.270657400e-06f);
    q = fmaf(q, s, 9.220430572e-05f);
    q = fmaf(q, s, -1.108776002e-03f);
    q = fmaf(q, s, 9.826695057e-03f);
    q = fmaf(q, s, -6.636358108e-02f);
    q = fmaf(q, s, 3.989096663e-01f);
    return x * fmaf(xc, q, 0.5f);
}
// The same two polynomials on PAIRS of elements (round 4): gfx950's v_pk_fma_f32 / v_pk_mul_f32 do two fp32 operations per lane in the issue
// slot of one, the constants ride along as scalar operands broadcast to both halves.  Same operations in the same order per element -
// bit-identical to the scalar forms above - at 21 instead of 38 VALU instructions per pair (alpha * acc + bias included): the fc1-forward
// epilogue computes GELU and GELU' of 128 values per lane and tile.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, float c) { return __builtin_elementwise_fma(a, b, (f32x2){c, c}); }
__device__ __forceinline__ f32x2 gelu_poly_grad2(f32x2 x, f32x2& dy) {
    const f32x2 xc = {__builtin_amdgcn_fmed3f(x.x, -3.5f, 3.5f), __builtin_amdgcn_fmed3f(x.y, -3.5f, 3.5f)};
    const f32x2 s = xc * xc;
    f32x2 q = {-2.815794181e-09f, -2.815794181e-09f}, dq = {-4.223691272e-08f, -4.223691272e-08f};
    q = pk_fma(q, s, 1.818798183e-07f);   dq = pk_fma(dq, s, 2.364437638e-06f);
    q = pk_fma(q, s, -5.270657400e-06f);  dq = pk_fma(dq, s, -5.797723140e-05f);
    q = pk_fma(q, s, 9.220430572e-05f);   dq = pk_fma(dq, s, 8.298387515e-04f);
    q = pk_fma(q, s, -1.108776002e-03f);  dq = pk_fma(dq, s, -7.761432013e-03f);
    q = pk_fma(q, s, 9.826695057e-03f);   dq = pk_fma(dq, s, 4.913347528e-02f);
    q = pk_fma(q, s, -6.636358108e-02f);  dq = pk_fma(dq, s, -1.990907432e-01f);
    q = pk_fma(q, s, 3.989096663e-01f);   dq = pk_fma(dq, s, 3.989096663e-01f);
    const f32x2 cdf = pk_fma(xc, q, 0.5f);
    dy = __builtin_elementwise_fma(xc, dq, cdf);
    return x * cdf;
}
__device__ __forceinline__ f32x2 gelu_poly2(f32x2 x) {
    const f32x2 xc = {__builtin_amdgcn_fmed3f(x.x, -3.5f, 3.5f), __builtin_amdgcn_fmed3f(x.y, -3.5f, 3.5f)};
    const f32x2 s = xc * xc;
    f32x2 q = {-2.815794181e-09f, -2.815794181e-09f};
    q = pk_fma(q, s, 1.818798183e-07f);
    q = pk_fma(q, s, -5.270657400e-06f);
    q = pk_fma(q, s, 9.220430572e-05f);
    q = pk_fma(q, s, -1.108776002e-03f);
    q = pk_fma(q, s, 9.826695057e-03f);
    q = pk_fma(q, s, -6.636358108e-02f);
    q = pk_fma(q, s, 3.989096663e-01f);
    return x * pk_fma(xc, q, 0.5f);
}
__device__ __forceinline__ float dgelu_erf(float x) {
    float g;
    const float cdf = norm_cdf(x, g);
    return cdf + x * 0.39894228040143268f * g;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// Full-wave sum through the DPP network (no LDS crossbar, no index registers): quad butterflies, half-row / row mirrors, then the two
// row broadcasts that gfx9-family DPP offers; the total arrives in lane 63 and is returned as a wave-uniform (scalar) value.
// All 64 lanes must be active.
__device__ __forceinline__ float wave_sum_uniform(float v) {
#define SS_DPP_ADD(ctrl, rmask) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rmask, 0xf, true))
    SS_DPP_ADD(0xB1, 0xf);      // quad_perm [1,0,3,2]
    SS_DPP_ADD(0x4E, 0xf);      // quad_perm [2,3,0,1]
    SS_DPP_ADD(0x141, 0xf);     // row_half_mirror
    SS_DPP_ADD(0x140, 0xf);     // row_mirror
    SS_DPP_ADD(0x142, 0xa);     // row_bcast15 -> rows 1, 3
    SS_DPP_ADD(0x143, 0xc);     // row_bcast31 -> rows 2, 3
#undef SS_DPP_ADD
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}

// counter-based RNG for dropout: one 32-bit hash per element index, reproducible in backward.  32-bit arithmetic only (the
// 64-bit splitmix used first cost ~35 VALU instructions per element - more than the GELU it sat next to): the index and
// seed halves are folded with odd multipliers, then the two-round "lowbias32" finaliser (bias < 0.11 bits / output bit).
__device__ __forceinline__ uint32_t hash_u32(uint64_t seed, uint64_t idx) {
    uint32_t x = (uint32_t)idx * 0x9E3779B1u + (uint32_t)(idx >> 32) * 0x85EBCA77u + (uint32_t)seed + (uint32_t)(seed >> 32) * 0xC2B2AE3Du;
    x ^= x >> 16; x *= 0x21F0AAADu;
    x ^= x >> 15; x *= 0x735A2D97u;
    x ^= x >> 15;
    return x;
}
// the same hash for an index known to fit 32 bits, with the seed folded once per kernel (seed_fold): one multiply-add
// plus the finaliser per element and no 64-bit temporaries (the attention backward kernels have no registers to spare)
__device__ __forceinline__ uint32_t seed_fold(uint64_t seed) { return (uint32_t)seed + (uint32_t)(seed >> 32) * 0xC2B2AE3Du; }
__device__ __forceinline__ bool dropout_keep32(uint32_t seedf, uint32_t idx, uint32_t thresh) {
    uint32_t x = idx * 0x9E3779B1u + seedf;
    x ^= x >> 16; x *= 0x21F0AAADu;
    x ^= x >> 15; x *= 0x735A2D97u;
    x ^= x >> 15;
    return x >= thresh;
}
// keep-probability test: keep iff hash >= p * 2^32
__device__ __forceinline__ bool dropout_keep(uint64_t seed, uint64_t idx, uint32_t thresh) {
    return hash_u32(seed, idx) >= thresh;
}

// XCD-aware, bijective block remap: XCD x (= bid % 8, observed dispatch) gets a contiguous chunk of tiles
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, x = bid & 7, j = bid >> 3;
    const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + j;
}
