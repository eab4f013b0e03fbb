// Issue-rate microbenchmark for the long-sequence attention design (round 6): what does a softmax instruction cost beside MFMAs on gfx950,
// at one / two / three waves per SIMD?   hipcc --offload-arch=gfx950 -O3 -o ubench_issue ubench_issue.hip && ./ubench_issue
// One workgroup on one CU; every wave times its own loop with s_memtime (shader cycles).  Prints cycles per loop body and per instruction.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

#define REP2(x) x x
#define REP4(x) REP2(x) REP2(x)
#define REP8(x) REP4(x) REP4(x)
#define REP16(x) REP8(x) REP8(x)

#define ITERS 2000

// mode: which body.  role split: `split` > 0 => waves with (wave / 4) odd run body B instead of body A (two waves per SIMD: wave w sits on
// SIMD w % 4, so waves 0-3 / 4-7 are the two residents of each SIMD)
template <int BODY>
__device__ __forceinline__ void body(float (&v)[16], f32x16& acc0, f32x16& acc1, bf16x8 a, bf16x8 b) {
    if constexpr (BODY == 0) {          // 16 independent v_exp_f32
        asm volatile(
            "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
            "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
            "v_exp_f32 %8, %8\n v_exp_f32 %9, %9\n v_exp_f32 %10, %10\n v_exp_f32 %11, %11\n"
            "v_exp_f32 %12, %12\n v_exp_f32 %13, %13\n v_exp_f32 %14, %14\n v_exp_f32 %15, %15\n"
            : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]),
              "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]));
    } else if constexpr (BODY == 1) {   // 16 independent v_fma_f32
        asm volatile(
            "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %4\n v_fma_f32 %3, %3, %4, %5\n"
            "v_fma_f32 %4, %4, %5, %6\n v_fma_f32 %5, %5, %6, %7\n v_fma_f32 %6, %6, %7, %8\n v_fma_f32 %7, %7, %8, %9\n"
            "v_fma_f32 %8, %8, %9, %10\n v_fma_f32 %9, %9, %10, %11\n v_fma_f32 %10, %10, %11, %12\n v_fma_f32 %11, %11, %12, %13\n"
            "v_fma_f32 %12, %12, %13, %14\n v_fma_f32 %13, %13, %14, %15\n v_fma_f32 %14, %14, %15, %0\n v_fma_f32 %15, %15, %0, %1\n"
            : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]),
              "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]));
    } else if constexpr (BODY == 2) {   // 16 v_max_f32 (two-operand VOP2)
        asm volatile(
            "v_max_f32 %0, %0, %1\n v_max_f32 %1, %1, %2\n v_max_f32 %2, %2, %3\n v_max_f32 %3, %3, %4\n"
            "v_max_f32 %4, %4, %5\n v_max_f32 %5, %5, %6\n v_max_f32 %6, %6, %7\n v_max_f32 %7, %7, %8\n"
            "v_max_f32 %8, %8, %9\n v_max_f32 %9, %9, %10\n v_max_f32 %10, %10, %11\n v_max_f32 %11, %11, %12\n"
            "v_max_f32 %12, %12, %13\n v_max_f32 %13, %13, %14\n v_max_f32 %14, %14, %15\n v_max_f32 %15, %15, %0\n"
            : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]),
              "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]));
    } else if constexpr (BODY == 3) {   // 8 v_cvt_pk_bf16_f32 + 8 v_pk_mul_f32 (packed fp32)
        asm volatile(
            "v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %4, %4, %5\n v_cvt_pk_bf16_f32 %6, %6, %7\n"
            "v_cvt_pk_bf16_f32 %8, %8, %9\n v_cvt_pk_bf16_f32 %10, %10, %11\n v_cvt_pk_bf16_f32 %12, %12, %13\n v_cvt_pk_bf16_f32 %14, %14, %15\n"
            "v_cvt_pk_bf16_f32 %1, %0, %1\n v_cvt_pk_bf16_f32 %3, %2, %3\n v_cvt_pk_bf16_f32 %5, %4, %5\n v_cvt_pk_bf16_f32 %7, %6, %7\n"
            "v_cvt_pk_bf16_f32 %9, %8, %9\n v_cvt_pk_bf16_f32 %11, %10, %11\n v_cvt_pk_bf16_f32 %13, %12, %13\n v_cvt_pk_bf16_f32 %15, %14, %15\n"
            : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]),
              "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]));
    } else if constexpr (BODY == 4) {   // 8 back-to-back MFMAs on two accumulators
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
        }
    }
}

// MFMA + K fillers of one kind per MFMA, all in ONE wave: 8 MFMAs per body.  KIND 0 = v_fma_f32, 1 = v_exp_f32, 2 = softmax mix
// (per MFMA: 2 exp, 2 fma, 2 max, 2 add, 1 cvt_pk = 9 instructions, the D = 64 attention ratio)
template <int K, int KIND>
__device__ __forceinline__ void body_mix(float (&v)[16], f32x16& acc0, f32x16& acc1, bf16x8 a, bf16x8 b) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (i & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
        else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
        if constexpr (KIND == 2) {
            asm volatile(
                "v_fma_f32 %0, %0, %8, %9\n v_exp_f32 %1, %1\n v_max_f32 %2, %2, %3\n v_add_f32 %4, %4, %5\n"
                "v_fma_f32 %5, %5, %8, %9\n v_exp_f32 %6, %6\n v_max_f32 %7, %7, %3\n v_add_f32 %3, %3, %5\n"
                "v_cvt_pk_bf16_f32 %2, %2, %7\n"
                : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])
                : "v"(v[8]), "v"(v[9]));
        } else {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k % 14]) : "v"(v[14]), "v"(v[15]));
                else asm volatile("v_exp_f32 %0, %0" : "+v"(v[k % 14]));
            }
        }
    }
}

struct Res { unsigned long long cyc; };

template <int BA, int BB>
__global__ __launch_bounds__(1024) void k_plain(Res* out, float seed, int split) {
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = seed + 0.001f * i * (threadIdx.x & 7);
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed * i); b[i] = (__bf16)(seed + i); }
    const int wave = threadIdx.x >> 6;
    const bool second = split && (((wave >> 2) & 1) != 0);
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (!second) {
#pragma unroll 1
        for (int it = 0; it < ITERS; ++it) body<BA>(v, acc0, acc1, a, b);
    } else {
#pragma unroll 1
        for (int it = 0; it < ITERS; ++it) body<BB>(v, acc0, acc1, a, b);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i] + acc0[i] + acc1[i];
    if (s == 12345.678f) out[1000].cyc = 1;
    if ((threadIdx.x & 63) == 0) out[wave].cyc = t1 - t0;
}

template <int K, int KIND>
__global__ __launch_bounds__(1024) void k_mix(Res* out, float seed) {
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = seed + 0.001f * i * (threadIdx.x & 7);
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed * i); b[i] = (__bf16)(seed + i); }
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < ITERS; ++it) body_mix<K, KIND>(v, acc0, acc1, a, b);
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i] + acc0[i] + acc1[i];
    if (s == 12345.678f) out[1000].cyc = 1;
    if ((threadIdx.x & 63) == 0) out[wave].cyc = t1 - t0;
}


// ---- the w64 attention tile as an instruction stream (round 6, second half): 32 MFMAs + 64 exp + 64 add + 32 cvt per tile and wave.
// SHAPE 0: the kernel's four phases (8 MFMAs alone | 8 x (MFMA + 4 exp + 4 add + 2 cvt) | the same | 8 MFMAs alone);
// SHAPE 1: uniform (32 x (MFMA + 2 exp + 2 add + 1 cvt)); SHAPE 2: two phases (16 MFMAs alone | 16 x (MFMA + 4 exp + 4 add + 2 cvt)).
// `rot`: waves 4-7 (the second resident of each SIMD) start half a tile later in the pattern.
// 0: 2 exp + 2 add + 1 cvt;  1: 2 exp + 1 cvt + 1 dot2;  2: 2 exp + 1 cvt;  3: 2 exp only;  4: 1 cvt + 2 add (no exp)
template <int MIX>
__device__ __forceinline__ void sm_unit_t(float (&v)[16], int k) {
    if constexpr (MIX == 0)
        asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_add_f32 %2, %2, %0\n v_add_f32 %3, %3, %1\n v_cvt_pk_bf16_f32 %4, %0, %1\n"
                     : "+v"(v[(2 * k) & 7]), "+v"(v[(2 * k + 1) & 7]), "+v"(v[8 + (k & 1)]), "+v"(v[10 + (k & 1)]), "+v"(v[12 + (k & 3)]));
    else if constexpr (MIX == 1)
        asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_cvt_pk_bf16_f32 %3, %0, %1\n v_dot2c_f32_bf16 %2, %3, %4\n"
                     : "+v"(v[(2 * k) & 7]), "+v"(v[(2 * k + 1) & 7]), "+v"(v[8 + (k & 1)]), "+v"(v[12 + (k & 3)]) : "v"(v[11]));
    else if constexpr (MIX == 2)
        asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_cvt_pk_bf16_f32 %2, %0, %1\n"
                     : "+v"(v[(2 * k) & 7]), "+v"(v[(2 * k + 1) & 7]), "+v"(v[12 + (k & 3)]));
    else if constexpr (MIX == 3)
        asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n" : "+v"(v[(2 * k) & 7]), "+v"(v[(2 * k + 1) & 7]));
    else
        asm volatile("v_add_f32 %2, %2, %0\n v_add_f32 %3, %3, %1\n v_cvt_pk_bf16_f32 %4, %0, %1\n"
                     : "+v"(v[(2 * k) & 7]), "+v"(v[(2 * k + 1) & 7]), "+v"(v[8 + (k & 1)]), "+v"(v[10 + (k & 1)]), "+v"(v[12 + (k & 3)]));
}
#ifndef SM_MIX
#define SM_MIX 0
#endif
__device__ __forceinline__ void sm_unit(float (&v)[16], int k) { sm_unit_t<SM_MIX>(v, k); }
template <int SHAPE>
__device__ __forceinline__ void tile_half(int half, float (&v)[16], f32x16 (&acc)[4], bf16x8 a, bf16x8 b) {
    if constexpr (SHAPE == 1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 3], 0, 0, 0);
            sm_unit(v, i);
        }
    } else if constexpr (SHAPE == 0) {
        if (half == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 3], 0, 0, 0);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 3], 0, 0, 0);
                sm_unit(v, 2 * i); sm_unit(v, 2 * i + 1);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 3], 0, 0, 0);
                sm_unit(v, 2 * i); sm_unit(v, 2 * i + 1);
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 3], 0, 0, 0);
        }
    } else {
        if (half == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 3], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 3], 0, 0, 0);
                sm_unit(v, 2 * i); sm_unit(v, 2 * i + 1);
            }
        }
    }
}

template <int SHAPE>
__global__ __launch_bounds__(1024) void k_tile(Res* out, float seed, int rot) {
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = seed + 0.001f * i * (threadIdx.x & 7);
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed * i); b[i] = (__bf16)(seed + i); }
    const int wave = threadIdx.x >> 6;
    const bool second = rot && (((wave >> 2) & 1) != 0);
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    unsigned long long t1 = t0;
    int done = 0;
    const unsigned long long span = 2000000ull;            // every wave runs for the same 2 M cycles: tiles completed while ALL co-run
    if (!second) {
#pragma unroll 1
        while (t1 - t0 < span) { tile_half<SHAPE>(0, v, acc, a, b); asm volatile("" ::: "memory"); tile_half<SHAPE>(1, v, acc, a, b); ++done; t1 = __builtin_readcyclecounter(); }
    } else {
#pragma unroll 1
        while (t1 - t0 < span) { tile_half<SHAPE>(1, v, acc, a, b); asm volatile("" ::: "memory"); tile_half<SHAPE>(0, v, acc, a, b); ++done; t1 = __builtin_readcyclecounter(); }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i] + acc[0][i] + acc[1][i] + acc[2][i] + acc[3][i];
    if (s == 12345.678f) out[1000].cyc = 1;
    if ((threadIdx.x & 63) == 0) { out[wave].cyc = t1 - t0; out[32 + wave].cyc = done; }
}

static Res* d_out;
static Res h_out[16];

template <typename F>
static void run(const char* name, F launch, int waves, int per_body, const char* unit) {
    hipMemset(d_out, 0, sizeof(Res) * 1024);
    launch(waves * 64);
    hipDeviceSynchronize();
    hipMemcpy(h_out, d_out, sizeof(Res) * 16, hipMemcpyDeviceToHost);
    printf("%-64s waves=%2d (%d/SIMD):", name, waves, (waves + 3) / 4);
    for (int w = 0; w < waves; w += (waves > 4 ? 4 : 1)) {
        const double cyc = (double)h_out[w].cyc / ITERS;
        printf("  w%d %.1f cyc/body = %.2f /%s", w, cyc, cyc / per_body, unit);
    }
    printf("\n");
}

template <typename F>
static void run_tiles(const char* name, F launch, int waves) {
    static Res h[64];
    hipMemset(d_out, 0, sizeof(Res) * 1024);
    launch(waves * 64);
    hipDeviceSynchronize();
    hipMemcpy(h, d_out, sizeof(Res) * 64, hipMemcpyDeviceToHost);
    double per_simd = 0;       // tiles per cycle on SIMD 0 (waves 0, 4, 8, 12)
    printf("%-64s waves=%2d (%d/SIMD):", name, waves, (waves + 3) / 4);
    for (int w = 0; w < waves; w += 4) {
        per_simd += (double)h[32 + w].cyc / (double)h[w].cyc;
        printf("  w%d %.0f cyc/tile", w, (double)h[w].cyc / (double)h[32 + w].cyc);
    }
    printf("  => %.0f cycles per tile and SIMD = %.3f of the matrix pipe (1024)\n", 1.0 / per_simd, 1024.0 * per_simd);
}

int main() {
    hipMalloc(&d_out, sizeof(Res) * 1024);
    // s_memtime unit check: cycles over a known wall time is not needed here - ratios between bodies on the same clock are what is read
    for (int waves : {4, 8, 12, 16}) {
        run("16 x v_exp_f32", [&](int n) { hipLaunchKernelGGL((k_plain<0, 0>), dim3(1), dim3(n), 0, 0, d_out, 0.5f, 0); }, waves, 16, "exp");
        run("16 x v_fma_f32", [&](int n) { hipLaunchKernelGGL((k_plain<1, 1>), dim3(1), dim3(n), 0, 0, d_out, 0.5f, 0); }, waves, 16, "fma");
        run("16 x v_max_f32", [&](int n) { hipLaunchKernelGGL((k_plain<2, 2>), dim3(1), dim3(n), 0, 0, d_out, 0.5f, 0); }, waves, 16, "max");
        run("16 x v_cvt_pk_bf16_f32", [&](int n) { hipLaunchKernelGGL((k_plain<3, 3>), dim3(1), dim3(n), 0, 0, d_out, 0.5f, 0); }, waves, 16, "cvt");
        run("8 x mfma_32x32x16_bf16", [&](int n) { hipLaunchKernelGGL((k_plain<4, 4>), dim3(1), dim3(n), 0, 0, d_out, 0.5f, 0); }, waves, 8, "mfma");
    }
    for (int waves : {4, 8, 12}) {
        run("1 wave: mfma + 4 fma", [&](int n) { hipLaunchKernelGGL((k_mix<4, 0>), dim3(1), dim3(n), 0, 0, d_out, 0.5f); }, waves, 8, "mfma");
        run("1 wave: mfma + 6 fma", [&](int n) { hipLaunchKernelGGL((k_mix<6, 0>), dim3(1), dim3(n), 0, 0, d_out, 0.5f); }, waves, 8, "mfma");
        run("1 wave: mfma + 8 fma", [&](int n) { hipLaunchKernelGGL((k_mix<8, 0>), dim3(1), dim3(n), 0, 0, d_out, 0.5f); }, waves, 8, "mfma");
        run("1 wave: mfma + 10 fma", [&](int n) { hipLaunchKernelGGL((k_mix<10, 0>), dim3(1), dim3(n), 0, 0, d_out, 0.5f); }, waves, 8, "mfma");
        run("1 wave: mfma + 12 fma", [&](int n) { hipLaunchKernelGGL((k_mix<12, 0>), dim3(1), dim3(n), 0, 0, d_out, 0.5f); }, waves, 8, "mfma");
        run("1 wave: mfma + 2 exp", [&](int n) { hipLaunchKernelGGL((k_mix<2, 1>), dim3(1), dim3(n), 0, 0, d_out, 0.5f); }, waves, 8, "mfma");
        run("1 wave: mfma + 4 exp", [&](int n) { hipLaunchKernelGGL((k_mix<4, 1>), dim3(1), dim3(n), 0, 0, d_out, 0.5f); }, waves, 8, "mfma");
        run("1 wave: mfma + softmax mix (9 instr)", [&](int n) { hipLaunchKernelGGL((k_mix<0, 2>), dim3(1), dim3(n), 0, 0, d_out, 0.5f); }, waves, 8, "mfma");
    }
    // two residents per SIMD with split roles: waves 0-3 MFMA only, waves 4-7 VALU only
    run("split: w0-3 mfma | w4-7 exp", [&](int n) { hipLaunchKernelGGL((k_plain<4, 0>), dim3(1), dim3(n), 0, 0, d_out, 0.5f, 1); }, 8, 8, "body-unit(8)");
    run("split: w0-3 mfma | w4-7 fma", [&](int n) { hipLaunchKernelGGL((k_plain<4, 1>), dim3(1), dim3(n), 0, 0, d_out, 0.5f, 1); }, 8, 8, "body-unit(8)");
    run("split: w0-3 exp  | w4-7 fma", [&](int n) { hipLaunchKernelGGL((k_plain<0, 1>), dim3(1), dim3(n), 0, 0, d_out, 0.5f, 1); }, 8, 16, "instr");
    // the attention tile as a stream: cycles per tile and wave; the matrix pipe needs 32 x 32 = 1024 per wave-tile
    printf("softmax unit per MFMA pair: mix %d\n", SM_MIX);
    for (int waves : {4, 8, 12, 16}) {
        for (int rot : {0, 1}) {
            if (waves == 4 && rot) continue;
            std::string tag = std::string(rot ? " (second resident half a tile behind)" : "");
            run_tiles(("tile, four phases" + tag).c_str(), [&](int n) { hipLaunchKernelGGL((k_tile<0>), dim3(1), dim3(n), 0, 0, d_out, 0.5f, rot); }, waves);
            run_tiles(("tile, uniform" + tag).c_str(), [&](int n) { hipLaunchKernelGGL((k_tile<1>), dim3(1), dim3(n), 0, 0, d_out, 0.5f, rot); }, waves);
            run_tiles(("tile, two phases" + tag).c_str(), [&](int n) { hipLaunchKernelGGL((k_tile<2>), dim3(1), dim3(n), 0, 0, d_out, 0.5f, rot); }, waves);
        }
    }
    hipFree(d_out);
    return 0;
}
