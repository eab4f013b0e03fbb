// MFMA GEMM with fused epilogues for the encoder hot path (K1, K4, K6-K8, K10, K13, K14, K16 of
// SURVEY.md section 2.2).  One kernel template:
//
//     C[M,N] = epilogue( alpha * opA(A)[M,K] . opB(B)[K,N] )
//
//   * T = float  : v_mfma_f32_32x32x2_f32  (exact fp32, parity mode), operands k-contiguous only
//   * T = bf16   : v_mfma_f32_32x32x16_bf16 (fp32 accumulate), operands k-contiguous or, for the
//                  backward GEMMs, stored transposed and fetched with ds_read_b64_tr_b16.
//
// Layout convention (all row-major):
//   TA = 0 : A stored [M,K] (k contiguous)          TA = 1 : A stored [K,M] (m contiguous)
//   TB = 0 : B stored [N,K] (k contiguous, = nn.Linear weight [out,in])
//   TB = 1 : B stored [K,N] (n contiguous)
//   forward  y  = x . W^T   -> TA=0, TB=0        (reference: nn.Linear inside timm Block / HF BertLayer)
//   dgrad    dx = dy . W    -> TA=0, TB=1
//   wgrad    dW = dy^T . x  -> TA=1, TB=1        (split-K, fp32 atomic accumulation)
//
// Block tile 128x128, 4 waves (2x2), each wave a 64x64 tile = 2x2 MFMA 32x32 accumulators.
// A K-tile is 128 bytes of k per row (64 bf16 / 32 fp32).  Operands are staged global -> VGPR -> LDS
// (16 B per lane per transfer) with the next tile's global loads in flight during the MFMAs.
// LDS rows are padded (144 B pitch for k-contiguous tiles, 320 B for transposed tiles) so that the
// ds_read_b128 / ds_read_b64_tr_b16 fragment fetches are bank-conflict free.
#include "common.h"

// Epilogue ablation switches of tools/dbg_fc1_epilogue.py (skip the GELU' store / the GELU itself / the output store through
// simseg_debug_gemm_stagger(1001..1003)) exist only in builds made with -DSS_GEMM_ABLATE: in the shipped library a left-over debug value
// cannot silently produce wrong activations, and simseg_debug_gemm_stagger refuses values >= 1000.
#ifdef SS_GEMM_ABLATE
#define SS_ABL(v) (p.stagger == (v))
#else
#define SS_ABL(v) false
#endif

namespace {

constexpr int BM = 128, BN = 128, NTHREADS = 256;
constexpr int MC_PITCH = 320;              // transposed tile (bf16): k-rows x (256 B + 64 B pad)
// KB = bytes of k per tile row (128 or 256).  k-contiguous tile: 128 rows x (KB + 16 B pad); transposed: KB/2 k-rows.
template <int KB> struct Geo {
    static constexpr int KC_PITCH = KB + 16;
    static constexpr int NCH = KB / 16;                       // 16-byte chunks per row
    static constexpr int LOADS = 128 * NCH / NTHREADS;        // 16-byte loads per thread per operand tile
    static constexpr int KROWS = KB / 2;                      // k-rows of a transposed bf16 tile
    static constexpr int TILE_BYTES = (128 * KC_PITCH > KROWS * MC_PITCH) ? 128 * KC_PITCH : KROWS * MC_PITCH;
    static constexpr int KSTEPS = KB / 32;                    // MFMA k-steps (32 B of k per lane-half pair)
};

struct GemmParams {
    const void* A; const void* B; void* C;
    int M, N, K;
    long lda, ldb, ldc;
    float alpha;
    const float* bias;       // [N] fp32 or null
    const float* rowscale;   // [M] fp32 or null  (applied before bias)
    const float* residual;   // fp32 [*, ldr] or null, added last
    long ldr;
    int act;                 // 0 none, 1 gelu(erf), 2 multiply by gelu'(aux)
    const void* aux;         // act==2: pre-activation, same dtype/ld as C
    void* aux_out;           // act==1: optional copy of the pre-activation, same dtype/ld as C
    int row_group;           // >0: output row r -> (r / G) * (G + 1) + 1 + r % G  (ViT token rows after [cls])
    int res_mod;             // residual row = 1 + r % G (pos_embed) instead of the output row
    int accumulate;          // C += result (fp32 output only; always set when split-K)
    unsigned long long* dbg_trace;   // debugging only (simseg_debug_gemm_trace): per block {start, K loop start, K loop end, end} wall-clock stamps + hardware id
    int stagger;             // ping-pong kernel: first-round blocks start (slot % 4) * stagger wall-clock ticks (10 ns) late (see launch_pp)
    int dbg_skip_epilogue;   // benchmarking only (simseg_set_gemm_variant(100 + v)): the accumulators are kept live but nothing is stored
    int ksplit;              // k-tiles per split-K slice
    int nsplit;              // number of split-K slices (grid = tiles * nsplit, slice-major so a slice's tiles share an XCD)
    // dropout on (acc*alpha + bias), before the residual:  keep iff hash(seed, row*N+col) >= thresh
    unsigned long long drop_seed; unsigned int drop_thresh; float drop_scale;
    float* colsum;           // optional [N]: += column sums of the stored output (bias gradient of the producing layer)
    int pf_next;             // ping-pong kernel: the tail's copies fetch the next tile of this XCD (see gemm_pp_kernel)
    int aux_blocked;         // act 3 / 4 on the ping-pong kernels: aux_out / aux is the tile-blocked accumulator image (simseg_gemm act codes 5 / 6)
};

template <typename T> struct TT;
template <> struct TT<float> { static constexpr int EPC = 4, BK = 32; };
template <> struct TT<bf16_t> { static constexpr int EPC = 8, BK = 64; };

// ---- global -> register staging ----------------------------------------------------------------
template <typename T, bool TRANS, bool AL, int KB>
__device__ __forceinline__ void load_tile(u32x4 (&r)[Geo<KB>::LOADS], const T* __restrict__ base, long ld, int row0, int lim,
                                          int k0, int K, int tid) {
    constexpr int EPC = TT<T>::EPC;
    constexpr int NCH = Geo<KB>::NCH;
#pragma unroll
    for (int i = 0; i < Geo<KB>::LOADS; ++i) {
        const int idx = tid + NTHREADS * i;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (!TRANS) {
            const int rr = row0 + idx / NCH, kk = k0 + (idx % NCH) * EPC;
            if constexpr (AL) {
                if (rr < lim && kk < K) v = *reinterpret_cast<const u32x4*>(base + (long)rr * ld + kk);
            } else {            // arbitrary K / leading dimension (fp32 only): element loads with a K bound
                static_assert(sizeof(T) == 4 || AL, "unaligned path is fp32 only");
                if (rr < lim) {
                    const T* src = base + (long)rr * ld + kk;
                    union { u32x4 q; T e[EPC]; } u;
                    u.q = v;
#pragma unroll
                    for (int j = 0; j < EPC; ++j)
                        if (kk + j < K) u.e[j] = src[j];
                    v = u.q;
                }
            }
        } else {
            const int kr = k0 + (idx >> 4), mm = row0 + (idx & 15) * 8;
            if (kr < K && mm < lim) v = *reinterpret_cast<const u32x4*>(base + (long)kr * ld + mm);
        }
        r[i] = v;
    }
}

template <bool TRANS, int KB>
__device__ __forceinline__ void store_tile(const u32x4 (&r)[Geo<KB>::LOADS], char* lds, int tid) {
    constexpr int NCH = Geo<KB>::NCH;
#pragma unroll
    for (int i = 0; i < Geo<KB>::LOADS; ++i) {
        const int idx = tid + NTHREADS * i;
        const int off = TRANS ? (idx >> 4) * MC_PITCH + (idx & 15) * 16 : (idx / NCH) * Geo<KB>::KC_PITCH + (idx % NCH) * 16;
        *reinterpret_cast<u32x4*>(lds + off) = r[i];
    }
}

// ---- LDS -> MFMA fragment ---------------------------------------------------------------------
// Returns the 16 bytes of k this lane feeds to the MFMA for tile row (row32 + lane%32), k-step kk.
template <bool TRANS, int KB>
__device__ __forceinline__ u32x4 read_frag(const char* lds, int row32, int kk, int lane) {
    if (!TRANS) {
        return *reinterpret_cast<const u32x4*>(lds + (row32 + (lane & 31)) * Geo<KB>::KC_PITCH + (2 * kk + (lane >> 5)) * 16);
    } else {
        // ds_read_b64_tr_b16: every 16-lane group fetches a [4 k][16 m] block; lane a supplies the address of
        // 4 consecutive m of k-row (a / 4) and receives the 4 k values of column a.
        const int a = lane & 15;
        const int m16 = row32 + ((lane >> 4) & 1) * 16;
        const int kb = kk * 16 + (lane >> 5) * 8;
        const char* p = lds + (kb + (a >> 2)) * MC_PITCH + (m16 + (a & 3) * 4) * 2;
        typedef s16x4 __attribute__((address_space(3))) * lptr;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(p));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(p + 4 * MC_PITCH));
        union { struct { s16x4 lo, hi; } s; u32x4 v; } u;
        u.s.lo = lo; u.s.hi = hi;
        return u.v;
    }
}

template <typename T>
__device__ __forceinline__ void mma(f32x16& acc, const u32x4& a, const u32x4& b) {
    if constexpr (sizeof(T) == 2) {
        union { u32x4 v; bf16x8 h; } ua, ub;
        ua.v = a; ub.v = b;
        acc = SS_MFMA_32x32x16(ua.h, ub.h, acc, 0, 0, 0);
    } else {
        union { u32x4 v; float f[4]; } ua, ub;
        ua.v = a; ub.v = b;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ua.f[j], ub.f[j], acc, 0, 0, 0);
    }
}

template <typename TO> __device__ __forceinline__ float ld_out(const TO* p) { return (float)*p; }
template <typename TO> __device__ __forceinline__ void st_out(TO* p, float v) { *p = (TO)v; }

// ---- epilogue -------------------------------------------------------------------------------------------------
// The 32x32 MFMA accumulator layout gives a lane ONE column and 16 scattered rows, so storing it directly means
// 2-byte (bf16) stores, 64 per lane: measured store-issue bound (a K=768 GEMM spent ~half its time there).  Instead each
// wave stages a 32-row x 64-column fp32 block of its accumulators in a private LDS scratch area and re-reads it
// row-major, 8 consecutive columns per lane: bias / GELU / residual / dropout run on 8-wide vectors and every global
// access (store, residual load, pre-activation load/save) is 16 bytes per lane, whole 128/256-byte row segments per
// 8 lanes.
constexpr int EP_PITCH = 68;                       // floats per staged row (64 + 4 pad)
constexpr int EP_WAVE_FLOATS = 32 * EP_PITCH;      // 8704 B per wave

template <typename TO> struct Vec8;
template <> struct Vec8<float> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[8]) {
        const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    static __device__ __forceinline__ void store(float* p, const float (&v)[8]) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
};
template <> struct Vec8<bf16_t> {
    static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
        union { u32x4 q; bf16x8 h; } u;
        u.q = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)u.h[e];
    }
    static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[8]) {
        union { u32x4 q; bf16x8 h; } u;
#pragma unroll
        for (int e = 0; e < 8; ++e) u.h[e] = (bf16_t)v[e];
        *reinterpret_cast<u32x4*>(p) = u.q;
    }
};

// Emits rows [row0, row0+32) x cols [col0, col0+64) from two 32x32 accumulators (left/right 32 columns).
// col1 = first output column of the RIGHT 32 staged columns (col0 + 32 for the contiguous tilings; the ping-pong kernel gives a wave
// two column blocks 128 apart).
template <typename TO>
__device__ __forceinline__ void epilogue_block(const GemmParams& p, const f32x16& accL, const f32x16& accR, float* wlds, int row0,
                                               int col0, int col1, int lane, bool atomic, bool vec_ok, float (&cs)[8]) {
    const int h2 = lane >> 5, cl = lane & 31;
    if (p.dbg_skip_epilogue) {         // debug: epilogue skipped (keeps the accumulators live), for fixed-cost attribution
        if (accL[0] + accR[0] == 12345.678f) static_cast<TO*>(p.C)[0] = (TO)1.f;
        return;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rr = (r & 3) + 8 * (r >> 2) + 4 * h2;
        wlds[rr * EP_PITCH + cl] = accL[r];
        wlds[rr * EP_PITCH + 32 + cl] = accR[r];
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    TO* C = static_cast<TO*>(p.C);
    const TO* aux = static_cast<const TO*>(p.aux);
    TO* aux_out = static_cast<TO*>(p.aux_out);
    if constexpr (sizeof(TO) == 4) {
        if (atomic) {      // split-K partial: plain alpha-scaled accumulate, one row of 64 consecutive floats per instruction
            const int col = lane < 32 ? col0 + lane : col1 + lane - 32;
            if (col < p.N) {
                for (int rr = 0; rr < 32; ++rr) {
                    const int row = row0 + rr;
                    if (row >= p.M) break;
                    atomicAdd(reinterpret_cast<float*>(C) + (long)row * p.ldc + col, wlds[rr * EP_PITCH + lane] * p.alpha);
                }
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            return;
        }
    }
    // deliberately NOT unrolled: the body is large (8-wide GELU / GELU' / dropout / residual variants) and an unrolled,
    // 4x-inlined epilogue overflowed the instruction cache (tens of microseconds of fetch stalls per tile, measured)
#pragma unroll 1
    for (int t = 0; t < 4; ++t) {
        const int q = lane + 64 * t;
        const int rr = q >> 3, c8 = (q & 7) * 8;
        const int row = row0 + rr, col = c8 < 32 ? col0 + c8 : col1 + c8 - 32;
        if (row >= p.M || col >= p.N) continue;
        float v[8];
        {
            const float4 a = *reinterpret_cast<const float4*>(wlds + rr * EP_PITCH + c8);
            const float4 b = *reinterpret_cast<const float4*>(wlds + rr * EP_PITCH + c8 + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        }
        const float rs = p.alpha * (p.rowscale ? p.rowscale[row] : 1.0f);
        long orow = row;
        if (p.row_group > 0) orow = (long)(row / p.row_group) * (p.row_group + 1) + 1 + row % p.row_group;
        const long o = orow * p.ldc + col;
        const long ro = p.residual ? (p.res_mod ? (long)(1 + row % p.row_group) : orow) * p.ldr + col : 0;
        const bool full = vec_ok && col + 8 <= p.N;
        if (full) {
            if (p.bias) {
                float b[8];
                Vec8<float>::load(p.bias + col, b);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] * rs + b[e];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= rs;
            }
            if (p.act == 1) {
                if (aux_out) Vec8<TO>::store(aux_out + o, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = gelu_erf(v[e]);
            } else if (p.act == 2) {
                float a[8];
                Vec8<TO>::load(aux + o, a);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= dgelu_erf(a[e]);
            } else if (p.act == 3) {      // GELU whose DERIVATIVE is saved (the CDF and the Gaussian are in registers anyway)
                float d[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = gelu_erf_grad(v[e], d[e]);
                if (aux_out) Vec8<TO>::store(aux_out + o, d);
            } else if (p.act == 4) {      // times the saved derivative: no transcendental work in the backward epilogue
                float a[8];
                Vec8<TO>::load(aux + o, a);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= a[e];
            }
            if (p.drop_thresh) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    v[e] = dropout_keep(p.drop_seed, (unsigned long long)row * p.N + col + e, p.drop_thresh) ? v[e] * p.drop_scale : 0.f;
            }
            if (p.residual) {
                float a[8];
                Vec8<float>::load(p.residual + ro, a);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += a[e];
            }
            if constexpr (sizeof(TO) == 4) {
                if (p.accumulate) {
                    float a[8];
                    Vec8<float>::load(reinterpret_cast<const float*>(C) + o, a);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += a[e];
                }
            }
            Vec8<TO>::store(C + o, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) cs[e] += v[e];
        } else {
#pragma unroll 1
            for (int e = 0; e < 8 && col + e < p.N; ++e) {      // edge / unaligned columns: rolled, element-wise
                float x = wlds[rr * EP_PITCH + c8 + e] * rs + (p.bias ? p.bias[col + e] : 0.f);
                if (p.act == 1) {
                    if (aux_out) st_out(aux_out + o + e, x);
                    x = gelu_erf(x);
                } else if (p.act == 2) {
                    x *= dgelu_erf(ld_out(aux + o + e));
                } else if (p.act == 3) {
                    float d;
                    x = gelu_erf_grad(x, d);
                    if (aux_out) st_out(aux_out + o + e, d);
                } else if (p.act == 4) {
                    x *= ld_out(aux + o + e);
                }
                if (p.drop_thresh) x = dropout_keep(p.drop_seed, (unsigned long long)row * p.N + col + e, p.drop_thresh) ? x * p.drop_scale : 0.f;
                if (p.residual) x += p.residual[ro + e];
                if constexpr (sizeof(TO) == 4) {
                    if (p.accumulate) x += ld_out(C + o + e);
                }
                st_out(C + o + e, x);
                if (p.colsum) atomicAdd(p.colsum + col + e, x);
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// lanes with equal lane%8 hold partial sums of the same 8 columns: fold the 8 row groups, then 8 lanes x 8 atomics
__device__ __forceinline__ void flush_colsum(const GemmParams& p, float (&cs)[8], int col0, int col1, int lane) {
    if (!p.colsum) return;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float v = cs[e];
        v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
        cs[e] = v;
    }
    if (lane < 8) {
        const int col = lane < 4 ? col0 + lane * 8 : col1 + (lane - 4) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (col + e < p.N) atomicAdd(p.colsum + col + e, cs[e]);
    }
}

// 16-byte vector accesses need aligned bases and leading dimensions
__device__ __forceinline__ bool epilogue_vec_ok(const GemmParams& p, int out_elem_bytes) {
    bool ok = (p.ldc % 8 == 0) && (((uintptr_t)p.C) % 16 == 0);
    if (p.residual) ok = ok && (p.ldr % 4 == 0) && (((uintptr_t)p.residual) % 16 == 0);
    if (p.bias) ok = ok && (((uintptr_t)p.bias) % 16 == 0);
    if (p.aux) ok = ok && (((uintptr_t)p.aux) % 16 == 0);
    if (p.aux_out) ok = ok && (((uintptr_t)p.aux_out) % 16 == 0);
    return ok;
}

template <typename T, typename TO, bool TA, bool TB, bool AL = true, int KB = 128>
__global__ __launch_bounds__(NTHREADS, KB == 128 ? 3 : 2) void gemm_kernel(GemmParams p) {
    constexpr int TILE_BYTES = Geo<KB>::TILE_BYTES;
    static_assert(2 * TILE_BYTES >= 4 * EP_WAVE_FLOATS * 4, "operand tiles double as epilogue scratch");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* ldsA = lds;
    char* ldsB = lds + TILE_BYTES;
    constexpr int BK = KB / (int)sizeof(T);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles = tiles_n * ((p.M + BM - 1) / BM);
    // XCD x (= blockIdx % 8) walks a contiguous range of (slice, tile) pairs, slice-major: the tiles of one split-K slice
    // run on one XCD and share its L2 copy of that slice's operand slabs
    const int q = xcd_remap(blockIdx.x, gridDim.x);
    const int kz = q / tiles, t = q - kz * tiles;
    const int m0 = (t / tiles_n) * BM, n0 = (t % tiles_n) * BN;

    const int nk = (p.K + BK - 1) / BK;
    const int kt0 = kz * p.ksplit;
    const int kt1 = min(nk, kt0 + p.ksplit);

    const T* A = static_cast<const T*>(p.A);
    const T* B = static_cast<const T*>(p.B);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    u32x4 ra[Geo<KB>::LOADS], rb[Geo<KB>::LOADS];
    if (kt0 < kt1) {
        load_tile<T, TA, AL, KB>(ra, A, p.lda, m0, p.M, kt0 * BK, p.K, tid);
        load_tile<T, TB, AL, KB>(rb, B, p.ldb, n0, p.N, kt0 * BK, p.K, tid);
    }
    for (int kt = kt0; kt < kt1; ++kt) {
        store_tile<TA, KB>(ra, ldsA, tid);
        store_tile<TB, KB>(rb, ldsB, tid);
        __syncthreads();
        if (kt + 1 < kt1) {
            load_tile<T, TA, AL, KB>(ra, A, p.lda, m0, p.M, (kt + 1) * BK, p.K, tid);
            load_tile<T, TB, AL, KB>(rb, B, p.ldb, n0, p.N, (kt + 1) * BK, p.K, tid);
        }
#pragma unroll
        for (int kk = 0; kk < Geo<KB>::KSTEPS; ++kk) {
            u32x4 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = read_frag<TA, KB>(ldsA, wm * 64 + i * 32, kk, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = read_frag<TB, KB>(ldsB, wn * 64 + j * 32, kk, lane);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mma<T>(acc[i][j], fa[i], fb[j]);
        }
        __syncthreads();
    }

    // ---- epilogue (the main loop's trailing barrier has released the operand tiles: reuse them as scratch)
    const bool atomic = p.nsplit > 1;
    const bool vec_ok = epilogue_vec_ok(p, sizeof(TO));
    float* wlds = reinterpret_cast<float*>(lds) + wave * EP_WAVE_FLOATS;
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int i = 0; i < 2; ++i) {         // rolled: one copy of the (large) epilogue body in the instruction stream
        f32x16 l = acc[0][0], r = acc[0][1];
        if (i == 1) { l = acc[1][0]; r = acc[1][1]; }
        epilogue_block<TO>(p, l, r, wlds, m0 + wm * 64 + i * 32, n0 + wn * 64, n0 + wn * 64 + 32, lane, atomic, vec_ok, cs);
    }
    flush_colsum(p, cs, n0 + wn * 64, n0 + wn * 64 + 32, lane);
}

// =================================================================================================================
// K14: dense patch x class-text similarity map with the row L2-normalisation fused in.
//   out[m, c] = < x[m,:] / max(||x[m,:]||, eps),  text[c,:] >        (tools/seg_evaluation.py:112 + :136 for every class)
// One block = 128 patch rows x ALL classes (C <= 32*NT <= 256), 4 waves of 32 rows each; the class matrix tile stays in
// LDS for the whole block, every A fragment is read once and feeds NT MFMAs, and the sum of squares of each row is
// accumulated from the very fragments that feed the MFMAs -- x is read exactly once from HBM.
// =================================================================================================================
template <typename T, int NT>
__global__ __launch_bounds__(NTHREADS) void simmap_kernel(const T* __restrict__ x, const T* __restrict__ text, float* __restrict__ out,
                                                          int M, int C, int K, float eps, int normalize) {
    constexpr int KB = 128;
    constexpr int BK = KB / (int)sizeof(T);
    constexpr int PITCH = Geo<KB>::KC_PITCH;
    constexpr int NB = (NT + 3) / 4;                  // 128-row slabs of the class tile
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* ldsA = lds;
    char* ldsB = lds + 128 * PITCH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h2 = lane >> 5;
    const int m0 = blockIdx.x * 128;
    f32x16 acc[NT];
    {
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = zero;
    }
    float ss = 0.f;
    u32x4 ra[4], rb[NB][4];
    const int nk = (K + BK - 1) / BK;
    load_tile<T, false, true, KB>(ra, x, K, m0, M, 0, K, tid);
#pragma unroll
    for (int b = 0; b < NB; ++b) load_tile<T, false, true, KB>(rb[b], text, K, b * 128, C, 0, K, tid);
    for (int kt = 0; kt < nk; ++kt) {
        store_tile<false, KB>(ra, ldsA, tid);
#pragma unroll
        for (int b = 0; b < NB; ++b) store_tile<false, KB>(rb[b], ldsB + b * 128 * PITCH, tid);
        __syncthreads();
        if (kt + 1 < nk) {
            load_tile<T, false, true, KB>(ra, x, K, m0, M, (kt + 1) * BK, K, tid);
#pragma unroll
            for (int b = 0; b < NB; ++b) load_tile<T, false, true, KB>(rb[b], text, K, b * 128, C, (kt + 1) * BK, K, tid);
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const u32x4 fa = read_frag<false, KB>(ldsA, wave * 32, kk, lane);
            if constexpr (sizeof(T) == 4) {
                union { u32x4 v; float f[4]; } u; u.v = fa;
                ss += u.f[0] * u.f[0] + u.f[1] * u.f[1] + u.f[2] * u.f[2] + u.f[3] * u.f[3];
            } else {
                union { u32x4 v; bf16x8 h; } u; u.v = fa;
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float f = (float)u.h[e]; ss += f * f; }
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                // A operand = patch rows (MFMA rows), B operand = class rows (MFMA columns): a lane owns ONE class column, so
                // every store instruction writes 32 consecutive classes of a row (128-byte runs)
                const u32x4 fb = read_frag<false, KB>(ldsB, j * 32, kk, lane);
                mma<T>(acc[j], fa, fb);
            }
        }
        __syncthreads();
    }
    ss += __shfl_xor(ss, 32, 64);
    const float rn = normalize ? 1.0f / fmaxf(sqrtf(ss), eps) : 1.0f;       // lane l (either half) holds the scale of patch row l % 32
    float rnv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) rnv[r] = __shfl(rn, (r & 3) + 8 * (r >> 2) + 4 * h2, 64);
    const int cl = lane & 31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2;
        if (row >= M) continue;
        float* orow = out + (long)row * C;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int c = j * 32 + cl;
            if (c < C) orow[c] = acc[j][r] * rnv[r];
        }
    }
}

template <typename T, int NT>
int launch_simmap(const T* x, const T* text, float* out, int M, int C, int K, float eps, int normalize, hipStream_t stream) {
    constexpr int SMEM = (128 + ((NT + 3) / 4) * 128) * Geo<128>::KC_PITCH;
    hipLaunchKernelGGL((simmap_kernel<T, NT>), dim3((M + 127) / 128), dim3(NTHREADS), SMEM, stream, x, text, out, M, C, K, eps, normalize);
    SS_LAUNCH_CHECK("simseg_patch_text_sim");
    return 0;
}

template <typename T>
int dispatch_simmap(const void* x, const void* text, float* out, int M, int C, int K, float eps, int normalize, hipStream_t s) {
    const T* a = static_cast<const T*>(x);
    const T* b = static_cast<const T*>(text);
    const int nt = (C + 31) / 32;
    switch (nt) {
        case 1: return launch_simmap<T, 1>(a, b, out, M, C, K, eps, normalize, s);
        case 2: return launch_simmap<T, 2>(a, b, out, M, C, K, eps, normalize, s);
        case 3: return launch_simmap<T, 3>(a, b, out, M, C, K, eps, normalize, s);
        case 4: return launch_simmap<T, 4>(a, b, out, M, C, K, eps, normalize, s);
        case 5: case 6: return launch_simmap<T, 6>(a, b, out, M, C, K, eps, normalize, s);
        default: return launch_simmap<T, 8>(a, b, out, M, C, K, eps, normalize, s);
    }
}

template <typename T, typename TO, bool TA, bool TB, bool AL = true, int KB = 128>
int launch(const GemmParams& p, int splitk, hipStream_t stream) {
    constexpr int BK = KB / (int)sizeof(T);
    constexpr int SMEM = 2 * Geo<KB>::TILE_BYTES;
    static bool configured = false;
    if (!configured && SMEM > 65536) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel<T, TO, TA, TB, AL, KB>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        if (e != hipSuccess) return simseg_set_error("simseg_gemm: cannot reserve %d bytes of LDS: %s", SMEM, hipGetErrorString(e));
        configured = true;
    }
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    const int nk = (p.K + BK - 1) / BK;
    GemmParams q = p;
    if (splitk < 1) splitk = 1;
    if (splitk > nk) splitk = nk > 0 ? nk : 1;
    q.ksplit = (nk + splitk - 1) / splitk;
    if (q.ksplit < 1) q.ksplit = 1;
    const int z = nk > 0 ? (nk + q.ksplit - 1) / q.ksplit : 1;
    q.nsplit = z;
    dim3 grid(tiles * z, 1, 1);
    hipLaunchKernelGGL((gemm_kernel<T, TO, TA, TB, AL, KB>), grid, dim3(NTHREADS), SMEM, stream, q);
    SS_LAUNCH_CHECK("simseg_gemm");
    return 0;
}


// =================================================================================================================
// Large-tile bf16 kernel: 256x256 block tile, 8 waves (2 x 4), each wave 128x64 = 4x2 MFMA 32x32 accumulators.
// Operand K-slabs go global -> LDS directly (global_load_lds_dwordx4, 1 KiB per wave-instruction, no VGPR staging) into
// a ring of NSTAGE slabs of BKE k-elements each, so NSTAGE-1 slabs are in flight while one is consumed; one raw
// s_barrier per slab and counted vmcnt waits (never a full drain in steady state).  The LDS image is lane-linear, so
// the bank-conflict swizzle is applied to the per-lane SOURCE address and undone on the fragment read:
//   k-contiguous slab  [256 rows][BKE*2 B]: 16-B slot s of row r holds k-chunk  s ^ f(r)
//   transposed slab    [BKE k-rows][512 B]: 16-B slot s of k-row kr holds m-chunk s ^ ((kr & 3) << 2)
// Requirements: bf16 operands, K % BKE == 0, 16-byte aligned contiguous extents (else the 128x128 kernel is used).
// =================================================================================================================

template <int BKE> __device__ __forceinline__ int kc_swz(int r) { return BKE == 64 ? ((r >> 1) & 7) : ((r >> 2) & 3); }

// issue the global->LDS copies of one operand slab (ROWS tile rows/cols x BKE k) for this wave
template <int BKE, bool TRANS, int ROWS, int NW>
__device__ __forceinline__ void glds_slab(const bf16_t* __restrict__ base, long ld, int row0, int lim, int k0, char* lds_slab,
                                          int wave, int lane) {
    constexpr int SLAB = ROWS * BKE * 2;
    constexpr int NI = SLAB / 1024 / NW;             // 1-KiB pieces per wave
    static_assert(NI >= 1, "slab too small for the block's waves");
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int piece = wave * NI + i;
        const bf16_t* src;
        if (!TRANS) {
            constexpr int SPR = BKE / 8;                 // 16-B slots per row
            constexpr int RPP = 64 / SPR;                // rows per 1-KiB piece
            const int r = piece * RPP + lane / SPR;
            const int c = (lane % SPR) ^ kc_swz<BKE>(r);
            int gr = row0 + r;
            gr = gr < lim ? gr : lim - 1;                // rows past the edge re-read the last row; never stored
            src = base + (long)gr * ld + k0 + c * 8;
        } else {
            constexpr int SPR = ROWS / 8;                // 16-B slots per k-row
            constexpr int RPP = 64 / SPR;
            const int kr = piece * RPP + lane / SPR;
            const int c = (lane % SPR) ^ ((kr & 3) << 2);
            int gc = row0 + c * 8;
            gc = gc < lim ? gc : lim - 8;
            src = base + (long)(k0 + kr) * ld + gc;
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds_slab + piece * 1024), 16, 0, 0);
    }
}

template <int BKE, bool TRANS, int ROWS>
__device__ __forceinline__ u32x4 read_frag_l(const char* slab, int row32, int kk, int lane) {
    if (!TRANS) {
        const int r = row32 + (lane & 31);
        const int c = 2 * kk + (lane >> 5);
        return *reinterpret_cast<const u32x4*>(slab + r * (BKE * 2) + ((c ^ kc_swz<BKE>(r)) << 4));
    } else {
        const int a = lane & 15;
        const int mm = row32 + ((lane >> 4) & 1) * 16 + (a & 3) * 4;
        const int kr = kk * 16 + (lane >> 5) * 8 + (a >> 2);
        const char* p = slab + kr * (ROWS * 2) + (((mm >> 3) ^ ((kr & 3) << 2)) << 4) + (mm & 7) * 2;
        typedef s16x4 __attribute__((address_space(3))) * lptr;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(p));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(p + 4 * (ROWS * 2)));
        union { struct { s16x4 lo, hi; } s; u32x4 v; } u;
        u.s.lo = lo; u.s.hi = hi;
        return u.v;
    }
}

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// TM x TN block tile, 8 waves as WM_ x WN_, each wave (TM/WM_) x (TN/WN_) = FM x FN MFMA 32x32 blocks.
template <typename TO, bool TA, bool TB, int BKE, int NSTAGE, int TM, int TN, int WM_, int WN_, int MINW>
__global__ __launch_bounds__(WM_ * WN_ * 64, MINW) void gemm_large_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int NW = WM_ * WN_;
    constexpr int SLAB_A = TM * BKE * 2, SLAB_B = TN * BKE * 2, STAGE = SLAB_A + SLAB_B;
    constexpr int LPS = SLAB_A / 1024 / NW + SLAB_B / 1024 / NW;    // global_load_lds per wave per stage
    constexpr int KSTEPS = BKE / 16;
    constexpr int FM = TM / WM_ / 32, FN = TN / WN_ / 32;
    static_assert(FN % 2 == 0, "the epilogue stages 64-column blocks");
    static_assert(NSTAGE * STAGE >= NW * EP_WAVE_FLOATS * 4, "operand ring doubles as epilogue scratch");
    const unsigned long long dbg_t0 = __builtin_readcyclecounter();
    unsigned long long dbg_t1 = 0, dbg_t2 = 0, dbg_t3 = 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN_, wn = wave % WN_;
    const int tiles_n = (p.N + TN - 1) / TN;
    const int tiles = tiles_n * ((p.M + TM - 1) / TM);
    const int q = xcd_remap(blockIdx.x, gridDim.x);
    const int kz = q / tiles, t = q - kz * tiles;
    const int m0 = (t / tiles_n) * TM, n0 = (t % tiles_n) * TN;
    const int nk = p.K / BKE;
    const int kt0 = kz * p.ksplit;
    const int kt1 = min(nk, kt0 + p.ksplit);
    const bf16_t* A = static_cast<const bf16_t*>(p.A);
    const bf16_t* B = static_cast<const bf16_t*>(p.B);

    f32x16 acc[FM][FN];
    {
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = zero;
    }
    // prologue: fill NSTAGE-1 stages
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s) {
        if (kt0 + s < kt1) {
            glds_slab<BKE, TA, TM, NW>(A, p.lda, m0, p.M, (kt0 + s) * BKE, lds + s * STAGE, wave, lane);
            glds_slab<BKE, TB, TN, NW>(B, p.ldb, n0, p.N, (kt0 + s) * BKE, lds + s * STAGE + SLAB_A, wave, lane);
        }
    }
    int stage = 0;
    dbg_t1 = __builtin_readcyclecounter();
    for (int kt = kt0; kt < kt1; ++kt) {
        // slab kt must have landed; up to NSTAGE-2 younger slabs may stay in flight
        const int ahead = min(NSTAGE - 2, kt1 - 1 - kt);
        if (NSTAGE >= 5 && ahead >= 3) wait_vm<3 * LPS>();
        else if (NSTAGE >= 4 && ahead >= 2) wait_vm<2 * LPS>();
        else if (NSTAGE >= 3 && ahead >= 1) wait_vm<LPS>();
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (kt == kt0) dbg_t2 = __builtin_readcyclecounter();
        const int nxt = kt + NSTAGE - 1;
        if (nxt < kt1) {
            int ns = stage + NSTAGE - 1; ns = ns >= NSTAGE ? ns - NSTAGE : ns;
            glds_slab<BKE, TA, TM, NW>(A, p.lda, m0, p.M, nxt * BKE, lds + ns * STAGE, wave, lane);
            glds_slab<BKE, TB, TN, NW>(B, p.ldb, n0, p.N, nxt * BKE, lds + ns * STAGE + SLAB_A, wave, lane);
        }
        const char* sa = lds + stage * STAGE;
        const char* sb = sa + SLAB_A;
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
            u32x4 fa[FM], fb[FN];
#pragma unroll
            for (int j = 0; j < FN; ++j) fb[j] = read_frag_l<BKE, TB, TN>(sb, wn * (FN * 32) + j * 32, kk, lane);
#pragma unroll
            for (int i = 0; i < FM; ++i) fa[i] = read_frag_l<BKE, TA, TM>(sa, wm * (FM * 32) + i * 32, kk, lane);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) mma<bf16_t>(acc[i][j], fa[i], fb[j]);
        }
        stage = stage + 1 == NSTAGE ? 0 : stage + 1;
    }
    const bool atomic = p.nsplit > 1;
    const bool vec_ok = epilogue_vec_ok(p, sizeof(TO));
    __builtin_amdgcn_s_barrier();                       // every wave is done with the operand ring: reuse it as scratch
    dbg_t3 = __builtin_readcyclecounter();
    float* wlds = reinterpret_cast<float*>(lds) + wave * EP_WAVE_FLOATS;
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int c = 0; c < FM * (FN / 2); ++c) {      // rolled: one copy of the (large) epilogue body in the instruction stream
        const int i = c / (FN / 2), jp = c % (FN / 2);
        f32x16 l = acc[0][0], r = acc[0][1];
#pragma unroll
        for (int cc = 1; cc < FM * (FN / 2); ++cc)
            if (cc == c) { l = acc[cc / (FN / 2)][2 * (cc % (FN / 2))]; r = acc[cc / (FN / 2)][2 * (cc % (FN / 2)) + 1]; }
        if (FN > 2 && jp == 0 && i > 0) { }       // (column sums are per 64-column block: flushed per block below)
        float csb[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const int cb = n0 + wn * (FN * 32) + jp * 64;
        epilogue_block<TO>(p, l, r, wlds, m0 + wm * (FM * 32) + i * 32, cb, cb + 32, lane, atomic, vec_ok, FN == 2 ? cs : csb);
        if (FN > 2) flush_colsum(p, csb, cb, cb + 32, lane);
    }
    if (FN == 2) flush_colsum(p, cs, n0 + wn * 64, n0 + wn * 64 + 32, lane);
    if (p.dbg_skip_epilogue && blockIdx.x == 0 && tid == 0) {    // debug timeline (cycle counter) of block 0 / wave 0
        unsigned long long* d = reinterpret_cast<unsigned long long*>(p.C);
        d[0] = dbg_t1 - dbg_t0; d[1] = dbg_t2 - dbg_t1; d[2] = dbg_t3 - dbg_t2; d[3] = __builtin_readcyclecounter() - dbg_t3;
    }
}

template <typename TO, bool TA, bool TB, int BKE, int NSTAGE, int TM, int TN, int WM_, int WN_, int MINW>
int launch_large(const GemmParams& p, int splitk, hipStream_t stream) {
    constexpr int SMEM = NSTAGE * (TM + TN) * BKE * 2;
    static bool configured = false;
    auto kern = gemm_large_kernel<TO, TA, TB, BKE, NSTAGE, TM, TN, WM_, WN_, MINW>;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        if (e != hipSuccess) return simseg_set_error("simseg_gemm: cannot reserve %d bytes of LDS: %s", SMEM, hipGetErrorString(e));
        configured = true;
    }
    const int tiles = ((p.M + TM - 1) / TM) * ((p.N + TN - 1) / TN);
    const int nk = p.K / BKE;
    GemmParams q = p;
    if (splitk < 1) splitk = 1;
    if (splitk > nk) splitk = nk;
    q.ksplit = (nk + splitk - 1) / splitk;
    const int z = (nk + q.ksplit - 1) / q.ksplit;
    q.nsplit = z;
    hipLaunchKernelGGL(kern, dim3(tiles * z, 1, 1), dim3(WM_ * WN_ * 64), SMEM, stream, q);
    SS_LAUNCH_CHECK("simseg_gemm(large)");
    return 0;
}



// =================================================================================================================
// Ping-pong ("8-phase") bf16 kernel: 256x256 block tile, K-tiles of 64, 8 waves in two groups of four.
//
// The 256x256 direct-to-LDS kernel above moves all eight waves through "wait for the slab - read fragments - MFMA" in lockstep:
// its K loop measures ~1000 TFLOP/s (MFMA pipe ~40 % busy; profiles/r1_gemm_vs_vendor.txt).  Here the two waves that share a SIMD
// (wave w and w+4 = the two wave rows) run half a phase apart: while one issues its LDS fragment reads and global->LDS copies,
// the other owns the matrix pipe with a cluster of 8 back-to-back MFMAs, and two workgroup barriers per phase keep the
// alternation strict (cdna_hip_programming.md, "The 256^2 8-phase template").
//
//   * wave (g, wn) owns output rows {h*128 + g*64 + [0,64)} and columns {h*128 + wn*32 + [0,32)}, h = 0,1: one 64x32 quadrant per
//     operand half, so every wave needs operand half 0 before half 1 and staging follows that order.
//   * a K-tile is four 16 KiB half-tiles (B half 0, A half 0, B half 1, A half 1 - the order the fragment reads need them),
//     staged one per phase with global_load_lds (2 per wave), PP_LEAD phases ahead of the phase that reads them, into a ring of
//     two K-tiles (128 KiB).  Counted vmcnt only; a half-tile is waited for one phase before it is read, ahead of that phase's
//     first barrier (both groups have then passed a barrier behind every wave's wait).
//   * per phase a wave reads 8 (an A half) or 4 (a B half) fragments and runs one quadrant x K=64 = 8 MFMA 32x32x16; fragment
//     registers: one A set (32) + two B sets (2 x 16) + 128 accumulators.
// =================================================================================================================
constexpr int PP_HALF = 128 * 64 * 2;
constexpr int PP_BREG = 4 * PP_HALF;           // LDS: [A: tile parity x half][B: tile parity x half][dummy], 16 KiB each
constexpr int PP_DUMMY = 8 * PP_HALF;

// per-lane byte offset (inside one operand half-tile's source region) of the 16 bytes this lane copies for 1-KiB piece `piece`
template <bool TRANS>
__device__ __forceinline__ unsigned pp_src_off(int piece, int lane, int row0, int lim, long ld) {
    if (!TRANS) {                               // [rows][K], k contiguous: 8 rows x 128 B per piece
        const int r = piece * 8 + (lane >> 3);
        const int c = (lane & 7) ^ kc_swz<64>(r);
        int gr = row0 + r;
        gr = gr < lim ? gr : lim - 1;           // rows past the edge re-read the last row; never stored
        return (unsigned)((long)gr * ld * 2 + c * 16);
    } else {                                    // [K][rows], row index contiguous: 4 k-rows x 256 B per piece
        const int kr = piece * 4 + (lane >> 4);
        const int c = (lane & 15) ^ ((kr & 3) << 2);
        int gc = row0 + c * 8;
        gc = gc < lim ? gc : lim - 8;
        return (unsigned)((long)kr * ld * 2 + gc * 2);
    }
}

// one global -> LDS copy (16 B per lane, 1 KiB per wave) in the scalar-base + 32-bit-lane-offset form: the per-lane state of a
// whole operand stream is ONE VGPR per piece (the builtin takes a 64-bit per-lane pointer: 2 VGPRs per piece plus the adds).
// M0 (LDS destination) is written in the same statement that uses it.  Not counted by the compiler: waits are explicit.
__device__ __forceinline__ void pp_glds16(unsigned voff, const char* sbase, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// Fragment addressing.  A half-tile is the slab image of read_frag_l<64, TRANS, 128>; its lane-dependent part is computed once:
//   k-contiguous: lane -> row (lane & 31), 16-byte chunk (2 kk + lane / 32) ^ swizzle(row): four addresses (one per kk); the
//                 32-row block index and the half-tile slot are immediate offsets (4096 and 16384 bytes)
//   transposed  : the swizzle depends on (lane, 32-row block) only: one address per 32-row block; kk and the second 4-k-row
//                 group are immediate offsets (4096 and 1024 bytes)
template <bool TRANS, int NB>
struct PPFrag {
    unsigned a[TRANS ? NB : 4];
    __device__ __forceinline__ void init(unsigned region, int row0, int lane) {
        if (!TRANS) {
            const int r = lane & 31;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                a[kk] = region + (row0 + r) * 128 + (((2 * kk + (lane >> 5)) ^ kc_swz<64>(r)) << 4);
        } else {
            const int q = lane & 15;
            const int kr = (lane >> 5) * 8 + (q >> 2);
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int mm = row0 + i * 32 + ((lane >> 4) & 1) * 16 + (q & 3) * 4;
                a[i] = region + kr * 256 + (((mm >> 3) ^ ((kr & 3) << 2)) << 4) + (mm & 7) * 2;
            }
        }
#pragma unroll
        for (int i = 0; i < (TRANS ? NB : 4); ++i) asm volatile("" : "+v"(a[i]));      // keep them as they are: base + immediate reads
    }
    __device__ __forceinline__ void pin() {
#pragma unroll
        for (int i = 0; i < (TRANS ? NB : 4); ++i) asm volatile("" : "+v"(a[i]));
    }
    __device__ __forceinline__ void toggle(unsigned bit) {
#pragma unroll
        for (int i = 0; i < (TRANS ? NB : 4); ++i) a[i] ^= bit;
    }
    // fragment of 32-row block i, k-step kk, from the half-tile at byte offset `slot` of this operand's region
    __device__ __forceinline__ u32x4 read(const char* lds, int slot, int i, int kk) const {
        if (!TRANS) {
            return *reinterpret_cast<const u32x4*>(lds + a[kk] + (slot + i * 4096));
        } else {
            typedef s16x4 __attribute__((address_space(3))) * lptr;
            const char* p = lds + a[i] + (slot + kk * 4096);
            s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(p));
            s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(p + 1024));
            union { struct { s16x4 lo, hi; } s; u32x4 v; } u;
            u.s.lo = lo; u.s.hi = hi;
            return u.v;
        }
    }
};

// ---- accumulator-layout epilogues of the 256x256 ping-pong kernels (shared by the per-tile and the persistent kernel) -------------
// bf16 output without a per-element loaded operand (qkv / fc1 forward, the plain dgrads - most of the step's GEMM launches):
// bias / GELU run on the accumulators in MFMA layout (a lane owns ONE column: the bias is a scalar per lane), pairs of rows are packed to
// bf16, staged TRANSPOSED ([column][row], 8-byte writes) in the wave's 4608-byte scratch `tl` and read back through ds_read_b64_tr_b16, which
// hands every lane 4 consecutive columns of one row: two reads = one 16-byte store.  ~70 instructions per 32x64 block.
// EK = the epilogue kinds compiled into an instantiation beyond the plain ones (act 0 / 1 / 3 with row-major tensors): 0 = none, 1 = times the
// saved derivative, row-major (act 4), 2 = GELU with the derivative saved as the tile-blocked accumulator image (act 5), 3 = times that image
// (act 6).  Each of the three sets the kernel's register high-water mark (32 column-sum partials and two operand sets for 1; prefetched 16-register
// operand sets for 3; a second converted accumulator block for 2): compiled into EVERY bf16-output kernel, state that lives across the K loop is
// parked in scratch around every tile of every launch (round 3 ended with 22-75 spilled VGPRs in all of them), so each kind has instantiations
// of its own and the plain kernels - most launches of the step - carry none of it (tests/test_host_logic.py reads .vgpr_spill_count).
// 8-bit image of the saved GELU' (simseg_gemm act 7 / 8): q = round((g + 0.132) * 255 / 1.264), g in [-0.1290, 1.1290]
constexpr float GELU8_SCALE = 255.0f / 1.264f, GELU8_OFF = 0.132f * (255.0f / 1.264f);

template <int EK = 1>
__device__ __forceinline__ void pp_epilogue_bf16(const GemmParams& p, const f32x16 (&acc)[4][2], char* tl, int m0, int c0, int c1, int grp, int lane) {
    constexpr int TP = 72;                                   // bytes per staged column (32 rows x 2 B + 8 B pad: conflict-free)
    const int cl = lane & 31, h2 = lane >> 5, g4 = lane >> 4, a16 = lane & 15;
    const float bL = p.bias ? p.bias[c0 + cl] : 0.f, bR = p.bias ? p.bias[c1 + cl] : 0.f;
    bf16_t* Cb = reinterpret_cast<bf16_t*>(p.C);
    bf16_t* Xb = reinterpret_cast<bf16_t*>(p.aux_out);
    typedef s16x4 __attribute__((address_space(3))) * lptr;
    auto emit = [&](const f32x16& xl, const f32x16& xr, bf16_t* dst, int row0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            union { bf16_t h[4]; uint2 u; } pl, pr;
#pragma unroll
            for (int e = 0; e < 4; ++e) { pl.h[e] = (bf16_t)xl[4 * q + e]; pr.h[e] = (bf16_t)xr[4 * q + e]; }
            *reinterpret_cast<uint2*>(tl + cl * TP + (8 * q + 4 * h2) * 2) = pl.u;
            *reinterpret_cast<uint2*>(tl + (32 + cl) * TP + (8 * q + 4 * h2) * 2) = pr.u;
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // this lane's four 8-column groups of its row: columns c0 + 8 (g4 >> 1) + {0, 16, 128, 144} (c1 = c0 + 128): one
        // address, immediate offsets
        bf16_t* drow = dst + (long)(row0 + (g4 & 1) * 16 + a16) * p.ldc + c0 + (g4 >> 1) * 8;
        const char* src0 = tl + ((g4 >> 1) * 8 + (a16 >> 2)) * TP + ((g4 & 1) * 16 + (a16 & 3) * 4) * 2;
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) {
            const char* src = src0 + sidx * 16 * TP;         // staged column octet (g4 >> 1) + 2 sidx: 0-3 left fragment, 4-7 right
            union { struct { s16x4 lo, hi; } s; u32x4 v; } u;
            u.s.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(src));
            u.s.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(src + 4 * TP));
            *reinterpret_cast<u32x4*>(drow + (sidx & 1) * 16 + (sidx >> 1) * 128) = u.v;
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    // Tile-blocked accumulator image of the saved GELU' (act codes 5 / 6 of simseg_gemm): the fc1 forward and the dgrad through fc2 are the
    // same M x N x K problem on the same tiling, and the derivative is computed - and consumed - with a lane owning a column and its
    // registers the rows.  Stored as it lies in the registers ([tile][wave slot][32-row block][column half][lane][16 rows], 32 bytes per
    // lane, 1 KiB per wave-instruction pair, fully coalesced) it needs no transposition on either side: the forward drops the second
    // trip through the staging scratch, the backward loads 32 bytes per lane straight into the accumulator layout, multiplies in fp32
    // (one rounding, exact column sums from two registers) and stores through the plain path.  Opaque to everything else: the tensor
    // is only ever handed from the one call to the other.
    const long blk0 = ((((long)(m0 >> 8) * (p.N >> 8) + (c0 >> 8)) * 8 + grp * 4 + ((c0 & 255) >> 5)) * 8) * 1024 + lane * 16;      // + (i * 2 + j) * 1024
    if (EK == 5 && p.act == 4 && p.aux_blocked == 2) {
        // act 8: times the 8-bit derivative image of act 7 (16 bytes per lane and 32-row block column: half the loads, half the operand registers)
        const unsigned char* A8 = reinterpret_cast<const unsigned char*>(p.aux) + blk0;
        u32x4 a8[2][2];                                      // [buffer][j]: the 16 rows of the left / right column, one byte each
        auto load8 = [&](int i, u32x4 (&dst)[2]) {
            dst[0] = *reinterpret_cast<const u32x4*>(A8 + (i * 2) * 1024);
            dst[1] = *reinterpret_cast<const u32x4*>(A8 + (i * 2 + 1) * 1024);
        };
        load8(0, a8[0]);
        f32x2 sLR = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < 3) load8(i + 1, a8[(i + 1) & 1]);
            f32x16 l, r;
            union { u32x4 v; unsigned w[4]; } gl, gr;
            gl.v = a8[i & 1][0]; gr.v = a8[i & 1][1];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const f32x2 q = {(float)((gl.w[e >> 2] >> (8 * (e & 3))) & 0xffu), (float)((gr.w[e >> 2] >> (8 * (e & 3))) & 0xffu)};      // v_cvt_f32_ubyteN
                const f32x2 gd = pk_fma(q, (f32x2){1.0f / GELU8_SCALE, 1.0f / GELU8_SCALE}, -GELU8_OFF / GELU8_SCALE);
                const f32x2 v = __builtin_elementwise_fma((f32x2){acc[i][0][e], acc[i][1][e]}, (f32x2){p.alpha, p.alpha}, (f32x2){bL, bR}) * gd;
                l[e] = v.x; r[e] = v.y;
                sLR += v;
            }
            emit(l, r, Cb, m0 + (i >> 1) * 128 + grp * 64 + (i & 1) * 32);
        }
        if (p.colsum) {
            float sL = sLR.x, sR = sLR.y;
            sL += __shfl_xor(sL, 32, 64); sR += __shfl_xor(sR, 32, 64);
            if (lane < 32) { atomicAdd(p.colsum + c0 + cl, sL); atomicAdd(p.colsum + c1 + cl, sR); }
        }
    } else if (EK == 3 && p.act == 4 && p.aux_blocked == 1) {
        const bf16_t* Ab = reinterpret_cast<const bf16_t*>(p.aux) + blk0;
        u32x4 ax[2][4];                                      // [buffer][j * 2 + half]: 16 rows of the left / right column
        // (one operand set requested right after the previous one is consumed, or scheduling barriers between the blocks, leave the
        //  register count where it is - 38 spilled in the persistent instantiation, 21 per-tile: it is the accumulators + the tile-walking
        //  state + ONE set that do not fit, and two sets keep a block's loads under the previous block's stores)
        auto load_blk = [&](int i, u32x4 (&dst)[4]) {
#pragma unroll
            for (int q = 0; q < 4; ++q) dst[q] = *reinterpret_cast<const u32x4*>(Ab + (i * 2 + (q >> 1)) * 1024 + (q & 1) * 8);
        };
        load_blk(0, ax[0]);
        f32x2 sLR = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < 3) load_blk(i + 1, ax[(i + 1) & 1]);
            f32x16 l, r;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                union { u32x4 v; bf16_t h[8]; } gl, gr;
                gl.v = ax[i & 1][q]; gr.v = ax[i & 1][2 + q];
#pragma unroll
                for (int e = 0; e < 8; ++e) {        // (left, right) as one pair: packed fp32 arithmetic
                    const f32x2 v = __builtin_elementwise_fma((f32x2){acc[i][0][8 * q + e], acc[i][1][8 * q + e]}, (f32x2){p.alpha, p.alpha}, (f32x2){bL, bR}) *
                                    (f32x2){(float)gl.h[e], (float)gr.h[e]};
                    l[8 * q + e] = v.x; r[8 * q + e] = v.y;
                    sLR += v;
                }
            }
            emit(l, r, Cb, m0 + (i >> 1) * 128 + grp * 64 + (i & 1) * 32);
        }
        if (p.colsum) {
            float sL = sLR.x, sR = sLR.y;
            sL += __shfl_xor(sL, 32, 64); sR += __shfl_xor(sR, 32, 64);
            if (lane < 32) { atomicAdd(p.colsum + c0 + cl, sL); atomicAdd(p.colsum + c1 + cl, sR); }
        }
    } else if (EK == 4 && p.act == 3 && Xb && p.aux_blocked == 2) {
        // act 7: the same image with ONE BYTE per element - GELU' lies in [-0.129, 1.129], stored as round((g + 0.132) * 255 / 1.264): a
        // uniform 0.005 grid, i.e. finer than a 16-bit float's spacing where most of the gradient's energy is (g in [0.5, 1.13]: bf16
        // steps of 0.004-0.008) and coarser only where |g| is small.  Relative RMS error of the product dY * g after its own rounding:
        // 2.7e-3 against 2.5e-3 with the 16-bit image (tests/test_gpu_kernels.py) - at half of the step's largest epilogue stream.
        unsigned char* B8 = reinterpret_cast<unsigned char*>(p.aux_out) + blk0;       // (blk0 counts elements: one byte each here)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x16 l, r;
            union { u32x4 v; unsigned w[4]; } ql, qr;
#pragma unroll
            for (int w4 = 0; w4 < 4; ++w4) {
                unsigned pl = 0, pr = 0;
#pragma unroll
                for (int b4 = 0; b4 < 4; ++b4) {
                    const int e = 4 * w4 + b4;
                    f32x2 d;
                    const f32x2 y = gelu_poly_grad2(__builtin_elementwise_fma((f32x2){acc[i][0][e], acc[i][1][e]}, (f32x2){p.alpha, p.alpha}, (f32x2){bL, bR}), d);
                    l[e] = y.x; r[e] = y.y;
                    const f32x2 t = pk_fma(d, (f32x2){GELU8_SCALE, GELU8_SCALE}, GELU8_OFF + 0.5f);      // truncation below = round to nearest
                    pl |= (unsigned)__builtin_amdgcn_fmed3f(t.x, 0.f, 255.f) << (8 * b4);      // (the clamp: a polynomial value a hair outside the range must not carry into the next byte)
                    pr |= (unsigned)__builtin_amdgcn_fmed3f(t.y, 0.f, 255.f) << (8 * b4);
                }
                ql.w[w4] = pl; qr.w[w4] = pr;
            }
            *reinterpret_cast<u32x4*>(B8 + (i * 2) * 1024) = ql.v;
            *reinterpret_cast<u32x4*>(B8 + (i * 2 + 1) * 1024) = qr.v;
            emit(l, r, Cb, m0 + (i >> 1) * 128 + grp * 64 + (i & 1) * 32);
        }
    } else if (EK == 2 && p.act == 3 && Xb && p.aux_blocked == 1) {
        bf16_t* Bb = Xb + blk0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x16 l, r;
            union { u32x4 v[2]; bf16_t h[16]; } dl, dr;
#pragma unroll
            for (int e = 0; e < 16; ++e) {           // the left / right element as one pair: packed fp32 arithmetic (common.h)
                f32x2 d;
                f32x2 y = __builtin_elementwise_fma((f32x2){acc[i][0][e], acc[i][1][e]}, (f32x2){p.alpha, p.alpha}, (f32x2){bL, bR});
                if (!SS_ABL(1002)) y = gelu_poly_grad2(y, d); else d = y;
                l[e] = y.x; r[e] = y.y;
                dl.h[e] = (bf16_t)d.x; dr.h[e] = (bf16_t)d.y;
            }
            if (!SS_ABL(1001))
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                *reinterpret_cast<u32x4*>(Bb + (i * 2) * 1024 + q * 8) = dl.v[q];
                *reinterpret_cast<u32x4*>(Bb + (i * 2 + 1) * 1024 + q * 8) = dr.v[q];
            }
            if (!SS_ABL(1003)) emit(l, r, Cb, m0 + (i >> 1) * 128 + grp * 64 + (i & 1) * 32);
        }
    } else if (EK == 1 && p.act == 4) {
        // times the saved derivative (the dgrad through fc2) + column sums (fc1's bias gradient).  The saved tensor is row-major like the
        // output, and `emit` hands every lane its results as 16-byte row pieces - the very pieces (same row, same eight columns) a 16-byte
        // load of the saved tensor returns.  So the product is formed THERE, on the way out: eight bf16 x bf16 -> fp32 products per piece,
        // rounded once more and stored.  No LDS trip for the saved tensor, no second pair of wave barriers per block (round 2 staged it
        // row-major, read it back transposed into the accumulator layout and multiplied before the packing: 18.2 us of epilogue per tile
        // against 4.2 us for the plain store - tools/dbg_gemm_trace.py).  The matmul result is rounded to the 16-bit type BEFORE the
        // multiplication, as torch's autocast does (a 16-bit matmul output feeding GELU's backward).  The loads of block i + 1 are requested
        // before the stores of block i.  Column sums: 32 per-lane partials (a lane's four column octets are the same in all four blocks),
        // reduced over the 32 lanes that share them through the wave's scratch.
        const bf16_t* Ab = reinterpret_cast<const bf16_t*>(p.aux);
        const int trow = (g4 & 1) * 16 + a16;
        const bf16_t* arow = Ab + (long)(m0 + grp * 64 + trow) * p.ldc + c0 + (g4 >> 1) * 8;
        u32x4 ax[2][4];              // (all sixteen pieces requested at once - the K loop's fragment registers are free - measured slower: 16.9 vs 14.5 us of epilogue per tile, spills)
        auto load_aux = [&](int i, u32x4 (&dst)[4]) {
            const bf16_t* a_i = arow + (long)((i >> 1) * 128 + (i & 1) * 32) * p.ldc;
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx) dst[sidx] = *reinterpret_cast<const u32x4*>(a_i + (sidx & 1) * 16 + (sidx >> 1) * 128);
        };
        load_aux(0, ax[0]);
        float cs[32];
#pragma unroll
        for (int e = 0; e < 32; ++e) cs[e] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < 3) load_aux(i + 1, ax[(i + 1) & 1]);
            const int row0 = m0 + (i >> 1) * 128 + grp * 64 + (i & 1) * 32;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                union { bf16_t h[4]; uint2 u; } pl, pr;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pl.h[e] = (bf16_t)(acc[i][0][4 * q + e] * p.alpha + bL);
                    pr.h[e] = (bf16_t)(acc[i][1][4 * q + e] * p.alpha + bR);
                }
                *reinterpret_cast<uint2*>(tl + cl * TP + (8 * q + 4 * h2) * 2) = pl.u;
                *reinterpret_cast<uint2*>(tl + (32 + cl) * TP + (8 * q + 4 * h2) * 2) = pr.u;
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            bf16_t* drow = Cb + (long)(row0 + (g4 & 1) * 16 + a16) * p.ldc + c0 + (g4 >> 1) * 8;
            const char* src0 = tl + ((g4 >> 1) * 8 + (a16 >> 2)) * TP + ((g4 & 1) * 16 + (a16 & 3) * 4) * 2;
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx) {
                const char* src = src0 + sidx * 16 * TP;
                union { struct { s16x4 lo, hi; } s; bf16_t h[8]; } u;
                u.s.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(src));
                u.s.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(src + 4 * TP));
                union { u32x4 v; bf16_t h[8]; } a, o;
                a.v = ax[i & 1][sidx];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float pv = (float)u.h[e] * (float)a.h[e];
                    cs[sidx * 8 + e] += pv;
                    o.h[e] = (bf16_t)pv;
                }
                *reinterpret_cast<u32x4*>(drow + (sidx & 1) * 16 + (sidx >> 1) * 128) = o.v;
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (p.colsum) {
            // lane L holds partials of columns c0 + (L >> 5) * 8 + (sidx & 1) * 16 + (sidx >> 1) * 128 + e, index sidx * 8 + e; the 32 lanes of a
            // half-wave hold the same 32 columns: [lane][32] floats through the scratch, lane L then sums index L & 31 over its half
            // (two passes of 16 indices: [lane][16] floats = 4 KiB, the scratch a wave owns in the persistent kernel is 4608 bytes)
            float* fs = reinterpret_cast<float*>(tl);
            const int idx = lane & 15, part = (lane >> 4) & 1, half = lane >> 5;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(fs + lane * 16 + 4 * q) = make_float4(cs[16 * t + 4 * q], cs[16 * t + 4 * q + 1], cs[16 * t + 4 * q + 2], cs[16 * t + 4 * q + 3]);
                __builtin_amdgcn_wave_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                float tot = 0.f;
#pragma unroll
                for (int l2 = 0; l2 < 16; ++l2) tot += fs[(half * 32 + part * 16 + l2) * 16 + idx];
                tot += __shfl_xor(tot, 16, 64);
                const int ci = 16 * t + idx, sidx = ci >> 3;
                if (part == 0) atomicAdd(p.colsum + c0 + half * 8 + (sidx & 1) * 16 + (sidx >> 1) * 128 + (ci & 7), tot);
                __builtin_amdgcn_wave_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
    } else if (p.act == 0) {       // no activation (qkv forward, the plain dgrads): a small body, unrolled - no accumulator selects
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x16 l, r;
#pragma unroll
            for (int e = 0; e < 16; ++e) { l[e] = acc[i][0][e] * p.alpha + bL; r[e] = acc[i][1][e] * p.alpha + bR; }
            emit(l, r, Cb, m0 + (i >> 1) * 128 + grp * 64 + (i & 1) * 32);
        }
    } else if (p.act == 3 && Xb) {     // the training forward of fc1: GELU stored, GELU' saved - unrolled (static accumulator indices)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x16 l, r, dl, dr;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                f32x2 d;
                const f32x2 y = gelu_poly_grad2(__builtin_elementwise_fma((f32x2){acc[i][0][e], acc[i][1][e]}, (f32x2){p.alpha, p.alpha}, (f32x2){bL, bR}), d);
                l[e] = y.x; r[e] = y.y;
                dl[e] = d.x; dr[e] = d.y;
            }
            const int row0 = m0 + (i >> 1) * 128 + grp * 64 + (i & 1) * 32;
            emit(l, r, Cb, row0);
            emit(dl, dr, Xb, row0);
        }
    } else
#pragma unroll
    for (int i = 0; i < 4; ++i) {          // (static accumulator indices: a rolled loop over the by-reference array ends up indexing it in scratch memory)
        f32x16 l = acc[i][0], r = acc[i][1];
        const int row0 = m0 + (i >> 1) * 128 + grp * 64 + (i & 1) * 32;
        f32x16 dl, dr;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float vl = l[e] * p.alpha + bL, vr = r[e] * p.alpha + bR;
            dl[e] = vl; dr[e] = vr;                          // act 1: the saved pre-activation
            if (p.act == 1) { const f32x2 y = gelu_poly2((f32x2){vl, vr}); vl = y.x; vr = y.y; }
            else if (p.act == 3) { f32x2 d; const f32x2 y = gelu_poly_grad2((f32x2){vl, vr}, d); vl = y.x; vr = y.y; dl[e] = d.x; dr[e] = d.y; }
            l[e] = vl; r[e] = vr;
        }
        emit(l, r, Cb, row0);
        if (p.act != 0 && Xb) emit(dl, dr, Xb, row0);
    }
}

// fp32 output (proj / fc2 forward: bias + dropout + fp32 residual): in the MFMA layout a lane owns one column and a store instruction covers
// 32 consecutive floats of a row per half-wave - whole 128-byte lines - so the accumulators are finished and stored where they are, the
// residual is loaded the same way, and nothing goes through LDS.
__device__ __forceinline__ void pp_epilogue_f32_direct(const GemmParams& p, const f32x16 (&acc)[4][2], int m0, int c0, int c1, int grp, int lane) {
    const int cl = lane & 31, h2 = lane >> 5;
    const float bL = p.bias ? p.bias[c0 + cl] : 0.f, bR = p.bias ? p.bias[c1 + cl] : 0.f;
    float* Cf = reinterpret_cast<float*>(p.C);
    f32x16 rl[2], rr[2];                                     // residual of the current / next fragment pair
    auto load_res = [&](int i, f32x16& xl, f32x16& xr) {
        const int row0 = m0 + (i >> 1) * 128 + grp * 64 + (i & 1) * 32;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const long ro = (long)(row0 + (e & 3) + 8 * (e >> 2) + 4 * h2) * p.ldr;
            xl[e] = __builtin_nontemporal_load(p.residual + ro + c0 + cl);       // the residual stream is read exactly once here
            xr[e] = __builtin_nontemporal_load(p.residual + ro + c1 + cl);
        }
    };
    if (p.residual) load_res(0, rl[0], rr[0]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (p.residual && i < 3) load_res(i + 1, rl[(i + 1) & 1], rr[(i + 1) & 1]);
        const int row0 = m0 + (i >> 1) * 128 + grp * 64 + (i & 1) * 32;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = row0 + (e & 3) + 8 * (e >> 2) + 4 * h2;
            float vl = acc[i][0][e] * p.alpha + bL, vr = acc[i][1][e] * p.alpha + bR;
            if (p.drop_thresh) {
                vl = dropout_keep(p.drop_seed, (unsigned long long)row * p.N + c0 + cl, p.drop_thresh) ? vl * p.drop_scale : 0.f;
                vr = dropout_keep(p.drop_seed, (unsigned long long)row * p.N + c1 + cl, p.drop_thresh) ? vr * p.drop_scale : 0.f;
            }
            if (p.residual) { vl += rl[i & 1][e]; vr += rr[i & 1][e]; }
            Cf[(long)row * p.ldc + c0 + cl] = vl;                         // (a non-temporal store here measured no different)
            Cf[(long)row * p.ldc + c1 + cl] = vr;
        }
    }
}

// PP_LEAD: half-tiles in flight ahead of the phase that reads them (3..5; the ring of two K-tiles allows up to 6).  PRIO: 1 = raise the
// wave priority around every MFMA cluster, 2 = static priority for the second group only, 0 = none.  Measured (tools/gemm_bench.py,
// r2): LEAD 3 / 4 / 5 and PRIO 0 / 1 / 2 are all within run-to-run noise (+-2 %) on the training shapes and on 4096^3 / 8192^3.
// BIG = 1 (round 3 experiment, variant 14): TWO phases per K-tile instead of four - 16 MFMAs between a phase's barriers.  Phase X reads A
// half 0 and both B halves (16 fragment reads), stages A half 1 of the next K-tile (2 copies) and multiplies A half 0 by both B halves;
// phase Y reads A half 1 (8 reads), stages B half 0 / A half 0 / B half 1 of the K-tile after next (6 copies: those slots were last read
// in phase X) and multiplies A half 1 by both B halves.  Every half-tile is requested two phases (one K-tile) before it is read;
// vmcnt(8) in every phase.  Half as many barrier pairs per MFMA.
template <typename TO, bool TA, bool TB, int PP_LEAD = 4, int PRIO = 1, int BIG = 0, int EK = 1>
__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wn = wave & 3;
    const int tiles_n = (p.N + 255) / 256;
    const int tiles = tiles_n * ((p.M + 255) / 256);
    const int q = xcd_remap(blockIdx.x, gridDim.x);
    const int kz = q / tiles, t = q - kz * tiles;
    const int m0 = (t / tiles_n) * 256, n0 = (t % tiles_n) * 256;
    const int nk = p.K / 64;
    const int kt0 = kz * p.ksplit;
    const int ntile = min(nk, kt0 + p.ksplit) - kt0;
    const int stot = 4 * ntile;                                      // half-tiles this block consumes
    // source of K-tile kt: base + kt * step (bytes); the per-lane parts are 32-bit offsets computed once
    const char* Ab = static_cast<const char*>(p.A) + (TA ? (long)kt0 * 64 * p.lda * 2 : (long)kt0 * 128);
    const char* Bb = static_cast<const char*>(p.B) + (TB ? (long)kt0 * 64 * p.ldb * 2 : (long)kt0 * 128);
    const long astep = TA ? 64 * p.lda * 2 : 128, bstep = TB ? 64 * p.ldb * 2 : 128;
    // The copies issued past the block's last K-tile (they keep the vmcnt arithmetic uniform) used to re-read the last K-tile into the
    // dummy slot.  They now read what the NEXT block of this XCD will want first - the first seven half-tiles of the tile 32 places on
    // in this XCD's run (one block per CU, 32 CUs per XCD: the tile that takes this round's place) - so that block's prologue finds its
    // operands in this XCD's L2 / the Infinity Cache instead of HBM.  A per-lane offset belongs to this tile's rows; the next tile's
    // rows are a scalar distance away as long as both tiles are full.  p.pf_next = 0 switches it off (A/B).
    long dA = 0, dB = 0;
    bool pf_ok = false;
    if (BIG && p.pf_next && p.nsplit == 1) {
        const int nb8 = (int)gridDim.x >> 3, rr8 = (int)gridDim.x & 7, xx = (int)blockIdx.x & 7, jj = (int)blockIdx.x >> 3;
        if (jj + 32 < nb8 + (xx < rr8 ? 1 : 0)) {
            const int tn_ = t + 32;                                  // (kz = 0: no split-K here)
            const int m1 = (tn_ / tiles_n) * 256, n1 = (tn_ % tiles_n) * 256;
            if (m0 + 256 <= p.M && n0 + 256 <= p.N && m1 + 256 <= p.M && n1 + 256 <= p.N) {
                pf_ok = true;
                dA = TA ? (long)(m1 - m0) * 2 : (long)(m1 - m0) * p.lda * 2;
                dB = TB ? (long)(n1 - n0) * 2 : (long)(n1 - n0) * p.ldb * 2;
            }
        }
    }
    unsigned offA[2][2], offB[2][2];                                 // [half][piece of this wave]
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            offA[h][i] = pp_src_off<TA>(wave * 2 + i, lane, m0 + h * 128, p.M, p.lda);
            offB[h][i] = pp_src_off<TB>(wave * 2 + i, lane, n0 + h * 128, p.N, p.ldb);
            asm volatile("" : "+v"(offA[h][i]), "+v"(offB[h][i]));
        }
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)lds;
    PPFrag<TA, 2> fra;
    PPFrag<TB, 1> frb;
    fra.init(0, grp * 64, lane);
    frb.init(PP_BREG, wn * 32, lane);

    f32x16 acc[4][2];
    {
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = zero;
    }
    // Half-tile h (in need order): K-tile h >> 2, KIND = h & 3 = {B half 0, A half 0, B half 1, A half 1}.  Every phase issues exactly
    // two copies per wave, so the vmcnt arithmetic is the same in every phase: past the last K-tile the copies re-read the last
    // K-tile into a dummy slot behind the ring.
#define PP_STAGE(H_, KIND)                                                                                            \
    {                                                                                                                 \
        const int h_ = (H_);                                                                                          \
        const bool live_ = h_ < stot;                                                                                 \
        const int kt_ = live_ ? (h_ >> 2) : (pf_ok ? ((h_ - stot) >> 2) : ntile - 1);                                 \
        const unsigned dst_ = lds0 + (live_ ? (((KIND) & 1) ? 0 : PP_BREG) + (((h_ >> 2) & 1) * 2 + ((KIND) >> 1)) * PP_HALF : PP_DUMMY) + \
                              wave * 2048;                                                                            \
        const char* sb_ = ((KIND) & 1) ? Ab + kt_ * astep + (live_ ? 0 : dA) : Bb + kt_ * bstep + (live_ ? 0 : dB);   \
        pp_glds16(((KIND) & 1) ? offA[(KIND) >> 1][0] : offB[(KIND) >> 1][0], sb_, dst_);                             \
        pp_glds16(((KIND) & 1) ? offA[(KIND) >> 1][1] : offB[(KIND) >> 1][1], sb_, dst_ + 1024);                      \
    }
    u32x4 fa[8], fb0[4], fb1[4];
#define PP_READ_A(PAR, HALF)                                                                       \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                               \
        _Pragma("unroll") for (int kk_ = 0; kk_ < 4; ++kk_) fa[i_ * 4 + kk_] = fra.read(lds, ((PAR) * 2 + (HALF)) * PP_HALF, i_, kk_);
#define PP_READ_B(FB, PAR, HALF) \
    _Pragma("unroll") for (int kk_ = 0; kk_ < 4; ++kk_) FB[kk_] = frb.read(lds, ((PAR) * 2 + (HALF)) * PP_HALF, 0, kk_);
#define PP_CLUSTER(RH, CH, FB)                                                                       \
    if (PRIO == 1) __builtin_amdgcn_s_setprio(1);                                                    \
    _Pragma("unroll") for (int kk_ = 0; kk_ < 4; ++kk_)                                              \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) mma<bf16_t>(acc[2 * (RH) + i_][CH], fa[i_ * 4 + kk_], FB[kk_]); \
    if (PRIO == 1) __builtin_amdgcn_s_setprio(0);

    if (p.stagger > 0 && blockIdx.x < 256) {                         // first round only: the phase offset then persists from round to round
        const unsigned long long until = wall_clock64() + (unsigned long long)(((blockIdx.x >> 3) & 3) * p.stagger);
        while (wall_clock64() < until) __builtin_amdgcn_s_sleep(8);
    }
    unsigned long long tr0 = 0, tr1 = 0, tr2 = 0;
    if (p.dbg_trace) tr0 = wall_clock64();
    if constexpr (BIG) {
        PP_STAGE(0, 0) PP_STAGE(1, 1) PP_STAGE(2, 2) PP_STAGE(3, 3) PP_STAGE(4, 0) PP_STAGE(5, 1) PP_STAGE(6, 2)
        wait_vm<8>();                                               // half-tiles 0..2 (what phase X of the first K-tile reads)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (grp == 1) __builtin_amdgcn_s_barrier();                 // the second group runs one barrier behind the first
        if (p.dbg_trace) tr1 = wall_clock64();
#define PP_BIGPHASE(READS, STAGES, RH, CA, FA_, CB, FB_)                                       \
    {                                                                                          \
        READS                                                                                  \
        STAGES                                                                                 \
        wait_vm<8>();                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        __builtin_amdgcn_s_barrier();                                                          \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                     \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        PP_CLUSTER(RH, CA, FA_)                                                                \
        PP_CLUSTER(RH, CB, FB_)                                                                \
        asm volatile("" : "+v"(acc[2 * (RH)][0]), "+v"(acc[2 * (RH) + 1][0]), "+v"(acc[2 * (RH)][1]), "+v"(acc[2 * (RH) + 1][1])); \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        __builtin_amdgcn_s_barrier();                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                     \
    }
        for (int tt = 0; tt < ntile; ++tt) {
            PP_BIGPHASE(PP_READ_B(fb0, 0, 0) PP_READ_A(0, 0) PP_READ_B(fb1, 0, 1), PP_STAGE(4 * tt + 7, 3), 0, 0, fb0, 1, fb1)
            frb.toggle(2 * PP_HALF);
            PP_BIGPHASE(PP_READ_A(0, 1), PP_STAGE(4 * tt + 8, 0) PP_STAGE(4 * tt + 9, 1) PP_STAGE(4 * tt + 10, 2), 1, 1, fb1, 0, fb0)
            fra.toggle(2 * PP_HALF);
        }
#undef PP_BIGPHASE
    } else {
    // ---- prologue: half-tiles 0..PP_LEAD on their way, 0 and 1 landed, B half 0 of the first K-tile in registers
    PP_STAGE(0, 0) PP_STAGE(1, 1) PP_STAGE(2, 2) PP_STAGE(3, 3)
    if (PP_LEAD >= 4) PP_STAGE(4, 0)
    if (PP_LEAD >= 5) PP_STAGE(5, 1)
    wait_vm<2 * (PP_LEAD - 1)>();                                   // half-tiles 0 and 1: the younger ones may be in flight
    if (PRIO == 2 && grp == 1) __builtin_amdgcn_s_setprio(1);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    PP_READ_B(fb0, 0, 0)
    if (grp == 1) __builtin_amdgcn_s_barrier();                     // the second group runs one barrier behind the first
    if (p.dbg_trace) tr1 = wall_clock64();

    // phase m = 4 * tile + q: reads half-tile m + 1, stages half-tile m + 1 + PP_LEAD and waits for half-tile m + 2 (the
    // PP_LEAD - 1 = 3 half-tiles staged after it may stay in flight)
#define PP_PHASE(M_, Q, READS, RH, CH, FB)                                                     \
    {                                                                                          \
        READS                                                                                  \
        PP_STAGE((M_) + 1 + PP_LEAD, ((Q) + 1 + PP_LEAD) & 3)                                  \
        wait_vm<2 * (PP_LEAD - 1)>();                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        __builtin_amdgcn_s_barrier();                                                          \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                     \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        PP_CLUSTER(RH, CH, FB)                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        __builtin_amdgcn_s_barrier();                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                     \
    }
    // One K-tile per trip.  The ring parity is a bit of the fragment addresses (slot = parity * 32 KiB + half * 16 KiB inside each
    // operand's 64 KiB region; the lane-dependent part stays below 16 KiB), toggled once per tile, so the loop body is the same
    // for every tile and each accumulator lives in one set of registers.  B half 0 of the NEXT tile is read into fb1 during the
    // fourth phase (fb1 is free after the third) and moved to fb0 behind that phase's MFMAs.
    for (int tt = 0; tt < ntile; ++tt) {
        PP_PHASE(4 * tt + 0, 0, PP_READ_A(0, 0), 0, 0, fb0)
        PP_PHASE(4 * tt + 1, 1, PP_READ_B(fb1, 0, 1), 0, 1, fb1)
        PP_PHASE(4 * tt + 2, 2, PP_READ_A(0, 1), 1, 1, fb1)
        frb.toggle(2 * PP_HALF);
        PP_PHASE(4 * tt + 3, 3, PP_READ_B(fb1, 0, 0), 1, 0, fb0)
        fra.toggle(2 * PP_HALF);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) fb0[kk] = fb1[kk];
    }
#undef PP_PHASE
    }
#undef PP_STAGE
#undef PP_READ_A
#undef PP_READ_B
#undef PP_CLUSTER
    // (the tail's copies land in the dummy slot, which no epilogue touches: when they fetch the next tile's operands - HBM latency -
    //  they are waited for at the END of the epilogue, counted behind its stores, instead of here)
    if (!pf_ok) wait_vm<0>();
    if (grp == 0) __builtin_amdgcn_s_barrier();                     // both groups have left the last phase: the ring is free
    if (p.dbg_trace) tr2 = wall_clock64();

    const bool atomic = p.nsplit > 1;
    const bool vec_ok = epilogue_vec_ok(p, sizeof(TO));
    float* wlds = reinterpret_cast<float*>(lds) + wave * EP_WAVE_FLOATS;
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int c0 = n0 + wn * 32, c1 = c0 + 128;
    if constexpr (sizeof(TO) == 2) {
        // bf16 output without a per-element loaded operand (qkv / fc1 forward, the plain dgrads - most of the step's GEMM launches):
        // the general epilogue below costs ~500 VALU / LDS instructions per wave and 32x64 block (fp32 staging, one ds_write_b32 per
        // element, row-major re-read, conversion) - 8-9 us of a 28 us K = 768 tile on the per-block timeline.  Here bias / GELU run
        // on the accumulators in MFMA layout (a lane owns ONE column: the bias is a scalar per lane), pairs of rows are packed to
        // bf16, staged TRANSPOSED ([column][row], 8-byte writes) and read back through ds_read_b64_tr_b16, which hands every lane 4
        // consecutive columns of one row: two reads = one 16-byte store.  ~70 instructions per block.
        const bool fastep = !atomic && vec_ok && !p.residual && !p.rowscale && p.row_group == 0 && !p.drop_thresh && !p.dbg_skip_epilogue &&
                            (p.act == 0 || p.act == 1 || (p.act == 3 && (!p.aux_blocked || (EK == 2 && p.aux_blocked == 1) || (EK == 4 && p.aux_blocked == 2))) ||
                             (((EK == 1 && !p.aux_blocked) || (EK == 3 && p.aux_blocked == 1) || (EK == 5 && p.aux_blocked == 2)) && p.act == 4 && p.aux)) && (p.act == 4 || !p.colsum) &&
                            m0 + 256 <= p.M && n0 + 256 <= p.N;
        if (fastep) {
            pp_epilogue_bf16<EK>(p, acc, reinterpret_cast<char*>(wlds), m0, c0, c1, grp, lane);
            if (pf_ok) wait_vm<16>();                                // >= 16 stores were issued after the tail's copies: those have landed
            if (p.dbg_trace && tid == 0) {
                const unsigned long long tr3 = wall_clock64();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                unsigned long long* d = p.dbg_trace + (long)blockIdx.x * 5;
                d[0] = tr0; d[1] = tr1; d[2] = tr2; d[3] = wall_clock64(); d[4] = tr3;
            }
            return;
        }
    }
    if constexpr (sizeof(TO) == 4) {
        // fp32 output (proj / fc2 forward: bias + dropout + fp32 residual): in the MFMA layout a lane owns one column and a store
        // instruction covers 32 consecutive floats of a row per half-wave - whole 128-byte lines - so the accumulators are finished and
        // stored where they are, the residual is loaded the same way, and nothing goes through LDS (the staged path: ~30 us of epilogue
        // per K = 768 tile for 256 KB read + 256 KB written).  Residual loads of fragment pair i + 1 are requested before the stores of
        // pair i, so no wait sits between a load and the stores ahead of it.
        const bool direct = !atomic && !p.rowscale && !p.aux && p.act == 0 && p.row_group == 0 && !p.colsum && !p.accumulate &&
                            !p.dbg_skip_epilogue && m0 + 256 <= p.M && n0 + 256 <= p.N;
        if (direct) {
            pp_epilogue_f32_direct(p, acc, m0, c0, c1, grp, lane);
            if (pf_ok) wait_vm<63>();                                // 128 stores behind the tail's copies
            if (p.dbg_trace && tid == 0) {
                const unsigned long long tr3 = wall_clock64();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                unsigned long long* d = p.dbg_trace + (long)blockIdx.x * 5;
                d[0] = tr0; d[1] = tr1; d[2] = tr2; d[3] = wall_clock64(); d[4] = tr3;
            }
            return;
        }
    }
#pragma unroll 1
    for (int i = 0; i < 4; ++i) {                                   // rolled: one copy of the (large) epilogue body
        f32x16 l = acc[0][0], r = acc[0][1];
#pragma unroll
        for (int ii = 1; ii < 4; ++ii)
            if (ii == i) { l = acc[ii][0]; r = acc[ii][1]; }
        if (p.dbg_trace && tid == 0) p.dbg_trace[(long)gridDim.x * 5 + (long)blockIdx.x * 4 + i] = wall_clock64();
        epilogue_block<TO>(p, l, r, wlds, m0 + (i >> 1) * 128 + grp * 64 + (i & 1) * 32, c0, c1, lane, atomic, vec_ok, cs);
    }
    flush_colsum(p, cs, c0, c1, lane);
    if (pf_ok) wait_vm<0>();
    if (p.dbg_trace && tid == 0) {
        const unsigned long long tr3 = wall_clock64();               // every store of this wave is issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // ... and acknowledged
        unsigned long long* d = p.dbg_trace + (long)blockIdx.x * 5;
        d[0] = tr0; d[1] = tr1; d[2] = tr2; d[3] = wall_clock64(); d[4] = tr3;
    }
}

template <typename TO, bool TA, bool TB, int LEAD = 4, int PRIO = 1, int BIG = 0, int EK = 1>
int launch_pp(const GemmParams& p, int splitk, hipStream_t stream) {
    constexpr int SMEM = 9 * PP_HALF;
    static bool configured = false;
    auto kern = gemm_pp_kernel<TO, TA, TB, LEAD, PRIO, BIG, EK>;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        if (e != hipSuccess) return simseg_set_error("simseg_gemm: cannot reserve %d bytes of LDS: %s", SMEM, hipGetErrorString(e));
        configured = true;
    }
    const int tiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
    const int nk = p.K / 64;
    GemmParams q = p;
    if (splitk < 1) splitk = 1;
    if (splitk > nk) splitk = nk;
    q.ksplit = (nk + splitk - 1) / splitk;
    const int z = (nk + q.ksplit - 1) / q.ksplit;
    q.nsplit = z;
    hipLaunchKernelGGL(kern, dim3(tiles * z, 1, 1), dim3(512), SMEM, stream, q);
    SS_LAUNCH_CHECK("simseg_gemm(ping-pong)");
    return 0;
}

// =================================================================================================================
// Persistent ping-pong kernel (round 3): one workgroup per CU walks a list of output tiles.
//
// What the per-tile kernel above loses between tiles (profiles/r2_gemm_shapes_and_tile_timeline.txt: K loop 19.0 us, prologue 2.2 us,
// epilogue 4.1 us per K = 768 tile, and the launch span is ~15 % longer than rounds x tile time): a new workgroup has to be dispatched,
// its first half-tiles fetched with nothing to do meanwhile, and all 256 CUs store their tiles at the same instants.  Here
//   * the operand stream never stops: the half-tile ring is fed across tile boundaries - while a tile's last K-tiles are consumed the
//     copies already fetch the NEXT tile's first half-tiles (the slots the per-tile kernel fills with dummy copies), so a tile's
//     "prologue" has happened under the previous tile's K loop and epilogue;
//   * the epilogue runs with those copies in flight / landed, on scratch taken from the ring slots of the K-tile consumed last
//     (free by construction: the next tile's first K-tile has the other parity, and of the K-tile after it only the B half 0 is staged);
//   * tiles are dealt out per XCD exactly as the per-tile kernel's dispatch order does (block b works for XCD b % 8 on that XCD's
//     contiguous chunk of tiles, stride = blocks per XCD), so the L2 sharing pattern is unchanged.
// Full tiles, no split-K, the two accumulator-layout epilogues only (the host checks); everything else stays on the per-tile kernel.
// =================================================================================================================
template <bool TRANS>
__device__ __forceinline__ unsigned pp2_lane_off(int piece, int lane, long ld) {
    if (!TRANS) {                               // [rows][K]: 8 rows x 128 B per 1-KiB piece, swizzled 16-byte chunk
        const int r = piece * 8 + (lane >> 3);
        const int c = (lane & 7) ^ kc_swz<64>(r);
        return (unsigned)((long)r * ld * 2 + c * 16);
    } else {                                    // [K][rows]: 4 k-rows x 256 B per piece
        const int kr = piece * 4 + (lane >> 4);
        const int c = (lane & 15) ^ ((kr & 3) << 2);
        return (unsigned)((long)kr * ld * 2 + c * 16);
    }
}

struct PP2Item { int m0, n0, ntile; const char* a0; const char* b0; };

template <bool TA, bool TB>
__device__ __forceinline__ PP2Item pp2_item(const GemmParams& p, int t, int tiles_n) {
    PP2Item it;
    it.m0 = (t / tiles_n) << 8;
    it.n0 = (t % tiles_n) << 8;
    it.ntile = p.K >> 6;
    it.a0 = static_cast<const char*>(p.A) + (TA ? (long)it.m0 * 2 : (long)it.m0 * p.lda * 2);
    it.b0 = static_cast<const char*>(p.B) + (TB ? (long)it.n0 * 2 : (long)it.n0 * p.ldb * 2);
    return it;
}

// SCHED (where a phase's two global->LDS copies are issued; the load segment of a phase - fragment reads + copies + waits - measured
// longer than the 8-MFMA segment it is supposed to hide under): 0 = both in the load segment (the per-tile kernel's order), 1 = both
// inside the wave's own MFMA cluster, 2 = one and one, 3 = both in the load segment but ahead of the fragment reads.
template <typename TO, bool TA, bool TB, int SCHED = 0, int EK = 1>
__global__ __launch_bounds__(512, 2) void gemm_pp2_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wn = wave & 3;
    const int tiles_n = p.N >> 8;
    const int total = tiles_n * (p.M >> 8);
    // XCD x = blockIdx % 8 owns a contiguous chunk of the tile list (xcd_remap's split); its blocks take every stride-th tile of it
    const int stride = (int)gridDim.x >> 3;
    const int x = blockIdx.x & 7;
    int ord = blockIdx.x >> 3;
    const int cq = total >> 3, cr = total & 7;
    const int cnt = cq + (x < cr ? 1 : 0);
    const int base = x < cr ? x * (cq + 1) : cr * (cq + 1) + (x - cr) * cq;
    if (ord >= cnt) return;                                          // (uniform per block)
    const long astep = TA ? 64 * p.lda * 2 : 128, bstep = TB ? 64 * p.ldb * 2 : 128;      // one K-tile further
    const long ahalf = TA ? 256 : 128 * p.lda * 2, bhalf = TB ? 256 : 128 * p.ldb * 2;    // operand half 1
    unsigned offA[2], offB[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        offA[i] = pp2_lane_off<TA>(wave * 2 + i, lane, p.lda);
        offB[i] = pp2_lane_off<TB>(wave * 2 + i, lane, p.ldb);
        asm volatile("" : "+v"(offA[i]), "+v"(offB[i]));
    }
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)lds;
    PPFrag<TA, 2> fra;
    PPFrag<TB, 1> frb;
    fra.init(0, grp * 64, lane);
    frb.init(PP_BREG, wn * 32, lane);

    PP2Item cur = pp2_item<TA, TB>(p, base + ord, tiles_n);
    bool has_next = ord + stride < cnt;
    PP2Item nxt = pp2_item<TA, TB>(p, base + (has_next ? ord + stride : ord), tiles_n);
    // staging state: the K-tile whose half-tiles are being requested
    const char* sA = cur.a0;
    const char* sB = cur.b0;
    int krem = cur.ntile - 1;                                        // K-tiles of the staged tile after this one
    bool nxt_ok = has_next;                                          // the next tile has not been entered by the staging yet
    unsigned spar = 0;                                               // ring parity of the staged K-tile
    bool sdummy = false;
    unsigned cpar = 0;                                               // ring parity of the K-tile being consumed

#define PP2_PIECE(KIND, I)                                                                                            \
    {                                                                                                                 \
        const char* sb_ = ((KIND) & 1) ? sA + ((KIND) >> 1) * ahalf : sB + ((KIND) >> 1) * bhalf;                     \
        const unsigned dst_ = lds0 + wave * 2048 + (I) * 1024 +                                                       \
                              (sdummy ? PP_DUMMY : (((KIND) & 1) ? 0 : PP_BREG) + (spar * 2 + ((KIND) >> 1)) * PP_HALF); \
        pp_glds16(((KIND) & 1) ? offA[I] : offB[I], sb_, dst_);                                                       \
    }
#define PP2_STAGE(KIND) PP2_PIECE(KIND, 0) PP2_PIECE(KIND, 1)
#define PP2_ADVANCE()                                                                    \
    {                                                                                    \
        if (krem > 0) { --krem; sA += astep; sB += bstep; }                              \
        else if (nxt_ok) { sA = nxt.a0; sB = nxt.b0; krem = nxt.ntile - 1; nxt_ok = false; } \
        else sdummy = true;                                                              \
        spar ^= 1u;                                                                      \
    }
    u32x4 fa[8], fb0[4], fb1[4];
#define PP_READ_A(HALF)                                                                            \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                               \
        _Pragma("unroll") for (int kk_ = 0; kk_ < 4; ++kk_) fa[i_ * 4 + kk_] = fra.read(lds, (HALF) * PP_HALF, i_, kk_);
#define PP_READ_B(FB, HALF) \
    _Pragma("unroll") for (int kk_ = 0; kk_ < 4; ++kk_) FB[kk_] = frb.read(lds, (HALF) * PP_HALF, 0, kk_);
#define PP_MMA(RH, CH, FB, KK, I) mma<bf16_t>(acc[2 * (RH) + (I)][CH], fa[(I) * 4 + (KK)], FB[KK]);
    // the wave group's 8 MFMAs of a phase; copies placed inside the cluster go behind the 2nd / 5th MFMA (the matrix pipe has work queued)
#define PP_CLUSTER(RH, CH, FB, KIND, PRE)                                                            \
    __builtin_amdgcn_s_setprio(1);                                                                   \
    PP_MMA(RH, CH, FB, 0, 0) PP_MMA(RH, CH, FB, 0, 1)                                                \
    if (SCHED == 1) { __builtin_amdgcn_sched_barrier(0); PRE PP2_PIECE(KIND, 0) __builtin_amdgcn_sched_barrier(0); } \
    PP_MMA(RH, CH, FB, 1, 0) PP_MMA(RH, CH, FB, 1, 1) PP_MMA(RH, CH, FB, 2, 0)                       \
    if (SCHED == 1 || SCHED == 2) { __builtin_amdgcn_sched_barrier(0); PP2_PIECE(KIND, 1) __builtin_amdgcn_sched_barrier(0); } \
    PP_MMA(RH, CH, FB, 2, 1) PP_MMA(RH, CH, FB, 3, 0) PP_MMA(RH, CH, FB, 3, 1)                       \
    /* the cluster's results are "used" here: without this the optimiser sinks the MFMAs (pure register operations) out of the */ \
    /* priority / barrier bracket into the next phase's load segment */                               \
    asm volatile("" : "+v"(acc[2 * (RH)][CH]), "+v"(acc[2 * (RH) + 1][CH]));                          \
    __builtin_amdgcn_s_setprio(0);
    // one phase: fragment reads of half-tile m + 1, two copies of half-tile m + 5, wait until half-tile m + 2 has landed (the
    // half-tiles requested after it may stay in flight), then this wave group's 8 MFMAs between two workgroup barriers.  PRE: the
    // staging state's step to the next K-tile, ahead of the first copy of a B half 0.
#define PP_PHASE(READS, KIND, PRE, RH, CH, FB)                                                 \
    {                                                                                          \
        if (SCHED == 3) { PRE PP2_STAGE(KIND) }                                                \
        READS                                                                                  \
        if (SCHED == 0) { PRE PP2_STAGE(KIND) }                                                \
        if (SCHED == 2) { PRE PP2_PIECE(KIND, 0) }                                             \
        wait_vm<SCHED == 1 ? 4 : (SCHED == 2 ? 5 : 6)>();                                      \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        __builtin_amdgcn_s_barrier();                                                          \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                     \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        PP_CLUSTER(RH, CH, FB, KIND, PRE)                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        __builtin_amdgcn_s_barrier();                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                     \
    }

    f32x16 acc[4][2];
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = zero;

    // ---- prologue (first tile only): half-tiles 0..4 requested
    PP2_STAGE(0) PP2_STAGE(1) PP2_STAGE(2) PP2_STAGE(3)
    PP2_ADVANCE()
    PP2_STAGE(0)

    for (;;) {
        // ---- top of a tile.  Its first five half-tiles were requested by the prologue / during the previous tile's last phases: all
        // landed after this wait (as are the previous epilogue's stores: vmcnt counts both); after the barrier every wave's copies are
        // visible and the epilogue's scratch slots are free again.  The wait is ALSO issued in its builtin form: the compiler's own
        // wait-count pass cannot see the inline-asm waits, and with loads / stores of the epilogue or register spills pending on a path
        // into the K loop it puts a vmcnt(0) in front of the loop's first fragment read, where it would drain the copy ring on every
        // trip.  The K loop's per-lane state is "used" first, so that anything the register allocator spilled around the epilogue is
        // reloaded ahead of the wait.
        asm volatile("" : "+v"(offA[0]), "+v"(offA[1]), "+v"(offB[0]), "+v"(offB[1]));
        fra.pin(); frb.pin();
        __builtin_amdgcn_s_waitcnt(0x0F70);                          // vmcnt(0)
        wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        PP_READ_B(fb0, 0)
        if (grp == 1) __builtin_amdgcn_s_barrier();                 // the second group runs one barrier behind the first
        for (int tt = 0; tt < cur.ntile; ++tt) {
            PP_PHASE(PP_READ_A(0), 1, , 0, 0, fb0)
            PP_PHASE(PP_READ_B(fb1, 1), 2, , 0, 1, fb1)
            PP_PHASE(PP_READ_A(1), 3, , 1, 1, fb1)
            frb.toggle(2 * PP_HALF);
            PP_PHASE(PP_READ_B(fb1, 0), 0, PP2_ADVANCE(), 1, 0, fb0)
            fra.toggle(2 * PP_HALF);
            cpar ^= 1u;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) fb0[kk] = fb1[kk];
        }
        if (grp == 0) __builtin_amdgcn_s_barrier();                 // both groups have left the tile's last phase
        __builtin_amdgcn_sched_barrier(0);
        // ---- epilogue.  Scratch: the ring slots of the K-tile consumed last (parity cpar ^ 1): A halves 0 / 1 (32 KiB, waves 0-6) and B
        // half 1 (wave 7); the copies in flight write the other parity and this parity's B half 0 only.
        const int c0 = cur.n0 + wn * 32, c1 = c0 + 128;
        if constexpr (sizeof(TO) == 2) {
            const unsigned lp = cpar ^ 1u;
            char* tl = lds + (wave < 7 ? lp * (2 * PP_HALF) + wave * 4608 : PP_BREG + (lp * 2 + 1) * PP_HALF);
            pp_epilogue_bf16<EK>(p, acc, tl, cur.m0, c0, c1, grp, lane);
        } else {
            pp_epilogue_f32_direct(p, acc, cur.m0, c0, c1, grp, lane);
        }
        if (!has_next) break;
        cur.m0 = nxt.m0; cur.n0 = nxt.n0; cur.ntile = nxt.ntile;
        ord += stride;
        has_next = ord + stride < cnt;
        nxt = pp2_item<TA, TB>(p, base + (has_next ? ord + stride : ord), tiles_n);
        nxt_ok = has_next;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = zero;
    }
    wait_vm<0>();                                                    // (dummy copies of the last tile's tail)
#undef PP_PHASE
#undef PP2_STAGE
#undef PP2_PIECE
#undef PP_MMA
#undef PP2_ADVANCE
#undef PP_READ_A
#undef PP_READ_B
#undef PP_CLUSTER
}

// host side: does this problem qualify for the persistent kernel (the conditions of its two epilogues, full tiles, more than one round)?
template <typename TO>
bool pp2_ok(const GemmParams& p, int splitk) {
    if (splitk > 1 || p.M % 256 || p.N % 256 || p.K % 64 || p.K < 128) return false;
    if (p.rowscale || p.row_group || p.dbg_skip_epilogue || p.dbg_trace) return false;
    const long tiles = (long)(p.M / 256) * (p.N / 256);
    if (tiles <= 256) return false;                                  // one round: nothing to overlap
    bool vec = (p.ldc % 8 == 0) && (((uintptr_t)p.C) % 16 == 0);
    if (p.bias) vec = vec && (((uintptr_t)p.bias) % 16 == 0);
    if (p.aux) vec = vec && (((uintptr_t)p.aux) % 16 == 0);
    if (p.aux_out) vec = vec && (((uintptr_t)p.aux_out) % 16 == 0);
    if (sizeof(TO) == 2)
        return vec && !p.residual && !p.drop_thresh && (p.act == 0 || p.act == 1 || p.act == 3 || (p.act == 4 && p.aux)) && (p.act == 4 || !p.colsum);
    return !p.aux && p.act == 0 && !p.colsum && !p.accumulate;
}

template <typename TO, bool TA, bool TB, int SCHED = 0, int EK = 1>
int launch_pp2(const GemmParams& p, hipStream_t stream, int reserve = 0) {
    constexpr int SMEM = 9 * PP_HALF;
    static bool configured = false;
    static int blocks = 0;
    auto kern = gemm_pp2_kernel<TO, TA, TB, SCHED, EK>;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        if (e != hipSuccess) return simseg_set_error("simseg_gemm: cannot reserve %d bytes of LDS: %s", SMEM, hipGetErrorString(e));
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8)
            return simseg_set_error("simseg_gemm: cannot read the CU count");
        blocks = cus & ~7;                                           // one workgroup per CU (144 KiB of LDS each), a multiple of the 8 XCDs
        configured = true;
    }
    GemmParams q = p;
    q.ksplit = p.K / 64; q.nsplit = 1;
    // reserve > 0 (variant 18): that many CUs are left to the kernels of the other stream - a persistent launch never hands a CU back, so
    // without a reserve the other tower's kernels wait for whole GEMM launches (the -6.7 % of round 3's first half)
    const int nb = reserve > 0 ? ((blocks - reserve) & ~7) : blocks;
    hipLaunchKernelGGL(kern, dim3(nb < 8 ? 8 : nb, 1, 1), dim3(512), SMEM, stream, q);
    SS_LAUNCH_CHECK("simseg_gemm(persistent ping-pong)");
    return 0;
}



// Debug / benchmarking selectors.  Thread-local: the entry points are otherwise stateless and re-entrant, and a selector set by a
// benchmark thread never changes what another caller's simseg_gemm launches.
thread_local int g_gemm_debug_skip_epilogue = 0;
thread_local unsigned long long* g_gemm_debug_trace = nullptr;
thread_local int g_gemm_stagger = 0;
thread_local int g_gemm_wgrad_blocks = 0;      // > 0: the split-K weight gradients use at most this many 256x256 blocks (simseg_debug_gemm_wgrad_blocks)
thread_local int g_gemm_last_variant = 0;      // 1 = 128x128 register-staged, 2 = 256x256 direct-to-LDS, 3 = 256x256 ping-pong
// variant: 0 = auto, 1 = 128x128 register-staged, 2 = 256x256 direct-to-LDS (BK64, 2 stages).  Other points of the design
// space were measured and dropped (profiles/r1_gemm_variants.txt): 256x256 with a 4-deep BK32 ring, 256x128 at 2 blocks/CU,
// 128x128 with BK=128, 256x128 with a 3-deep BK32 ring at 2 blocks/CU (563 vs 741 TFLOP/s aggregate), a persistent 256x256 kernel with
// cross-tile prefetch, an LDS-free epilogue on transposed accumulators (8-byte stores straight from registers: same time), a persistent
// one-block-per-CU kernel that defers a tile's stores into the first eight K-slabs of the next tile (NT +1 %, NN slower), non-temporal
// 16-byte output stores (same time; non-temporal 8-byte stores 30-50 % slower), four waves of 128x128 per block as the vendor library does (launch_large<..., 2, 2, 1>: 256 VGPR + 256 AGPR, no
// spills, correct, but 556 vs 673 TFLOP/s aggregate with compiler scheduling at one wave per SIMD), 256x128 tiles on four 128x64 waves at two
// blocks per CU so that one block's epilogue overlaps the other's K loop (launch_large<..., 32, 2|3, 256, 128, 2, 2, 2>: 630 vs 670), delaying the first round's blocks by 1/4..3/4 of a tile so the CUs' store bursts do not coincide (-1..-6 %), deeper BK32 rings (4 and 5 stages) and a
// two-group ping-pong schedule of the 256x256 kernel (MFMA phase of one wave per SIMD against the load phase of the other).

// =================================================================================================================
// Small-problem kernel: the reference tool's batch-1 call pattern (tools/seg_evaluation.py:84-85, 109: one image per forward ->
// GEMMs of 325 / 1025 rows).  A 128x128 tiling makes 9-54 blocks of such a problem - most of the 256 CUs idle, and every block's
// K loop a chain of exposed global-load latencies (profiles/r2_batch1_latency.txt).  Here:
//   * block tile (32 WM) x 64 with WM x WK = 4 waves: WM wave rows, and the WK wave groups SPLIT each K slab's four MFMA k-steps
//     between them (intra-block split-K, reduced through LDS before the epilogue), so a 1025 x 768 problem is 204 blocks
//     (WM = 2) and a 325 x 384 one 66 (WM = 1) with four waves busy on every CU that has a block;
//   * operands row.row (x . W^T), k-contiguous, streamed straight into LDS (global_load_lds, 16 B per lane) through a 4-stage
//     ring of 128-byte-deep slabs - three slabs in flight per block, counted vmcnt waits, one barrier per slab - for bf16 and
//     fp32 alike (a slab is 64 bf16 or 32 fp32 of k; the fragment layout is the 128x128 kernel's);
//   * the same fused epilogue (epilogue_block) on the WK = 0 waves.
// Requirements: no transposed operand, K * sizeof(T) % 128 == 0, 16-byte aligned rows, no split-K.
// =================================================================================================================
template <typename T, typename TO, int WM, int WK>
__global__ __launch_bounds__(256) void gemm_small_kernel(GemmParams p) {
    constexpr int TM = 32 * WM, TN = 64, NSTAGE = 4;
    constexpr int PA = TM / 8, PB = TN / 8;          // 1-KiB pieces (8 rows x 128 B) of the A / B slab
    constexpr int PPW = (PA + PB) / 4;               // pieces per wave per stage
    constexpr int STAGE = (TM + TN) * 128;
    constexpr int KPW = 4 / WK;                      // MFMA k-steps per wave per slab
    static_assert(WM * WK == 4 && (PA + PB) % 4 == 0, "four waves");
    static_assert(NSTAGE * STAGE >= (WK - 1) * WM * 8192 + WM * EP_WAVE_FLOATS * 4, "ring doubles as reduction + epilogue scratch");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wk = wave / WM;
    const int tiles_n = (p.N + TN - 1) / TN;
    const int t = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (t / tiles_n) * TM, n0 = (t % tiles_n) * TN;
    const int nk = (int)((long)p.K * (long)sizeof(T) / 128);

    // this wave's PPW pieces of every slab: per-lane source address (k = 0) and LDS offset within a stage
    const char* src[PPW];
    int dst[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int pi = wave * PPW + i;
        const bool isb = pi >= PA;
        const int pr = isb ? pi - PA : pi;
        const int r = pr * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);           // bank swizzle on the source chunk (the LDS image is lane-linear)
        int gr = (isb ? n0 : m0) + r;
        const int lim = isb ? p.N : p.M;
        gr = gr < lim ? gr : lim - 1;                        // rows past the edge re-read the last row; never stored
        src[i] = static_cast<const char*>(isb ? p.B : p.A) + ((long)gr * (isb ? p.ldb : p.lda)) * (long)sizeof(T) + c * 16;
        dst[i] = (isb ? TM * 128 : 0) + pr * 1024;
    }
    auto issue = [&](int stage, int kt) {
#pragma unroll
        for (int i = 0; i < PPW; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (long)kt * 128),
                                             (__attribute__((address_space(3))) void*)(lds + stage * STAGE + dst[i]), 16, 0, 0);
    };
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s)
        if (s < nk) issue(s, s);
    int stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const int ahead = min(NSTAGE - 2, nk - 1 - kt);      // younger slabs that may stay in flight
        if (ahead >= 2) wait_vm<2 * PPW>();
        else if (ahead == 1) wait_vm<PPW>();
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (kt + NSTAGE - 1 < nk) {
            int ns = stage + NSTAGE - 1; ns = ns >= NSTAGE ? ns - NSTAGE : ns;
            issue(ns, kt + NSTAGE - 1);
        }
        const char* sa = lds + stage * STAGE;
        const char* sb = sa + TM * 128;
#pragma unroll
        for (int kq = 0; kq < KPW; ++kq) {
            const int c = 2 * (wk * KPW + kq) + (lane >> 5);
            const int ra = wm * 32 + (lane & 31), rb = lane & 31;
            const u32x4 fa = *reinterpret_cast<const u32x4*>(sa + ra * 128 + ((c ^ ((ra >> 1) & 7)) << 4));
            const u32x4 fb0 = *reinterpret_cast<const u32x4*>(sb + rb * 128 + ((c ^ ((rb >> 1) & 7)) << 4));
            const u32x4 fb1 = *reinterpret_cast<const u32x4*>(sb + (rb + 32) * 128 + ((c ^ (((rb + 32) >> 1) & 7)) << 4));
            mma<T>(acc0, fa, fb0);
            mma<T>(acc1, fa, fb1);
        }
        stage = stage + 1 == NSTAGE ? 0 : stage + 1;
    }
    __builtin_amdgcn_s_barrier();                            // every wave is done with the ring
    if (WK > 1) {
        float* red = reinterpret_cast<float*>(lds);          // [wk - 1][wm][32 values][64 lanes]
        if (wk > 0) {
            float* dstp = red + ((wk - 1) * WM + wm) * 2048 + lane;
#pragma unroll
            for (int r = 0; r < 16; ++r) { dstp[r * 64] = acc0[r]; dstp[(16 + r) * 64] = acc1[r]; }
        }
        __syncthreads();
        if (wk > 0) return;
#pragma unroll
        for (int g = 0; g < WK - 1; ++g) {
            const float* sp = red + (g * WM + wm) * 2048 + lane;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] += sp[r * 64]; acc1[r] += sp[(16 + r) * 64]; }
        }
    }
    float* wlds = reinterpret_cast<float*>(lds + (WK - 1) * WM * 8192) + wm * EP_WAVE_FLOATS;
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    epilogue_block<TO>(p, acc0, acc1, wlds, m0 + wm * 32, n0, n0 + 32, lane, false, epilogue_vec_ok(p, sizeof(TO)), cs);
    flush_colsum(p, cs, n0, n0 + 32, lane);
}

template <typename T, typename TO, int WM, int WK>
int launch_small(const GemmParams& p, hipStream_t stream) {
    constexpr int SMEM = 4 * (32 * WM + 64) * 128;
    static bool configured = false;
    auto kern = gemm_small_kernel<T, TO, WM, WK>;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        if (e != hipSuccess) return simseg_set_error("simseg_gemm: cannot reserve %d bytes of LDS: %s", SMEM, hipGetErrorString(e));
        configured = true;
    }
    GemmParams q = p;
    q.ksplit = 0; q.nsplit = 1;
    const int tiles = ((p.M + 32 * WM - 1) / (32 * WM)) * ((p.N + 63) / 64);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), SMEM, stream, q);
    SS_LAUNCH_CHECK("simseg_gemm(small)");
    return 0;
}

// Problems that leave the 128x128 tiling with fewer blocks than CUs go to the small-problem kernel: 64x64 tiles when those
// already make ~a block per CU, 32x64 otherwise.
template <typename T, typename TO>
int dispatch_small(const GemmParams& p, hipStream_t s) {
    const long t64 = (long)((p.M + 63) / 64) * ((p.N + 63) / 64);
    if (t64 >= 192) return launch_small<T, TO, 2, 2>(p, s);
    return launch_small<T, TO, 1, 4>(p, s);
}

thread_local int g_gemm_variant = 0;

// the persistent kernel's instantiation for this call's epilogue kind (see pp_epilogue_bf16: each heavy kind has kernels of its own)
template <typename TO, bool TA, bool TB>
int launch_pp2_ek(const GemmParams& p, hipStream_t s, int reserve) {
    if constexpr (sizeof(TO) == 2) {      // (the kinds exist for 16-bit outputs only; the blocked image: fc1 forward = NT, dgrad through fc2 = NN)
        if (p.act == 4 && !p.aux_blocked) return launch_pp2<TO, TA, TB, 0, 1>(p, s, reserve);
        if constexpr (!TB) { if (p.act == 3 && p.aux_blocked == 1) return launch_pp2<TO, TA, TB, 0, 2>(p, s, reserve); }
        if constexpr (TB) { if (p.act == 4 && p.aux_blocked == 1) return launch_pp2<TO, TA, TB, 0, 3>(p, s, reserve); }
        if constexpr (!TB) { if (p.act == 3 && p.aux_blocked == 2) return launch_pp2<TO, TA, TB, 0, 4>(p, s, reserve); }
        if constexpr (TB) { if (p.act == 4 && p.aux_blocked == 2) return launch_pp2<TO, TA, TB, 0, 5>(p, s, reserve); }
    }
    if (p.aux_blocked) return simseg_set_error("simseg_gemm: act 5 is a forward (x . W^T) epilogue, act 6 a dgrad (d . W) epilogue, both with 16-bit outputs");
    return launch_pp2<TO, TA, TB, 0, 0>(p, s, reserve);
}

template <typename TO, bool TA, bool TB>
int dispatch_bf16(const GemmParams& p, int splitk, bool aligned, hipStream_t s) {
    const bool big_ok = aligned && p.K % 64 == 0 && p.M >= 256 && p.N >= 128;
    const bool four_phase = g_gemm_variant == 15;       // 15 = the automatic choice, ping-pong kernel on its four-phase schedule (A/B runs)
    int v = (four_phase || g_gemm_variant == 16 || g_gemm_variant == 18) ? 0 : g_gemm_variant;
    const int nk64 = p.K / 64;
    const int kper = (nk64 + (splitk > 1 ? splitk : 1) - 1) / (splitk > 1 ? splitk : 1);   // 64-deep slabs per block
    // measured (profiles/r1_gemm_variants.txt, after the epilogue was rolled to fit the instruction cache): the 256x256
    // direct-to-LDS kernel wins for k-contiguous A from K >= 768 on when there is at least one tile per CU; split-K wgrad
    // (transposed A) and small problems stay on the 128x128 kernel (3 blocks/CU, more blocks to spread)
    const int tiles256 = ((p.M + 255) / 256) * ((p.N + 255) / 256);
    if constexpr (TA && sizeof(TO) == 4) {
        // split-K weight gradient (accumulating into a zero-filled fp32 output): the caller's slice count targets the 128x128
        // kernel; for the 256x256 kernel pick the largest count that still fits ONE round of 256 blocks (a 288-block launch
        // runs two rounds, the second 12 % full)
        if ((v == 0 || v == 6 || v == 3) && v != 7 && big_ok && p.accumulate && splitk > 1 && p.M % 256 == 0 && p.N % 256 == 0 && tiles256 <= 128) {
            int sk = 256 / tiles256;
            static const int env_blocks = getenv("SIMSEG_GEMM_WGRAD_BLOCKS") ? atoi(getenv("SIMSEG_GEMM_WGRAD_BLOCKS")) : 0;      // (process-wide: backward runs on autograd's threads)
            const int wb = g_gemm_wgrad_blocks > 0 ? g_gemm_wgrad_blocks : env_blocks;
            if (wb > 0) sk = wb / tiles256 > 0 ? wb / tiles256 : 1;      // (CU-partition experiments)
            if (nk64 / sk < 16 && nk64 >= 64) sk = nk64 / 16;        // short contraction (packed text rows): fewer, 16-deep slices
            if (nk64 / sk >= 16 && tiles256 * sk >= 96 && (v == 3 || v == 0)) {
                g_gemm_last_variant = 3;
                return four_phase ? launch_pp<TO, TA, TB>(p, sk, s) : launch_pp<TO, TA, TB, 4, 1, 1>(p, sk, s);
            }
            if (nk64 / sk >= 16 && v == 6) { g_gemm_last_variant = 2; return launch_large<TO, TA, TB, 64, 2, 256, 256, 2, 4, 2>(p, sk, s); }
        }
        if (v == 3) v = 0;
        if (v == 6) v = 0;
        if (v == 7) v = 1;
    }
    // (measured, tools/gemm_mid_bench.py: the ping-pong kernel is ahead of the 128x128 one from ~96 of its tiles on - e.g. the packed text
    //  tower's ~22 k x 768 problems, 255 tiles: 34 vs 52 us at K = 768, 98 vs 151 us at K = 3072)
    // Round 5 (tools/gemm_vits_bench.py, the ViT-S forward shapes at M = 262400): a SHORT contraction (K = 384: six slabs) is still 22-26 %
    // faster on the ping-pong kernel once the problem is many rounds of tiles (qkv 502 -> 373 us, fc1 671 -> 521 us); and a column count that
    // leaves the last 256-wide tile mostly empty (N = 384: a third of the matrix work wasted) is better off on the 128x128 kernel up to K ~ 2500
    // (fc2 at K = 1536: 653 -> 543 us; from K = 3072 on the long K loop wins again: 223 vs 177 us at M = 64575).
    if (v == 0) {
        const int pad_n = ((p.N + 255) / 256) * 256 - p.N;
        const bool n_fits = pad_n * 6 <= p.N || kper >= 40;               // <= 1/6 of the columns are padding - or a contraction long enough
                                                                            // (K >= 2560: the split-bf16 form of an fp32 product) to make up for it
        v = (big_ok && !TA && tiles256 >= 96 && n_fits && (kper >= 12 || (kper >= 6 && tiles256 >= 1024))) ? 3 : 1;
    }
    if (v >= 10 && v <= 13 && !(big_ok && !TA)) v = 1;
    // one round of 128x128 tiles (more than the small-problem kernel takes, at most a block per CU): the ring variant's three slabs in
    // flight beat the register-staged kernel's one (14.7 vs 16.8 us on 1025 x 2304 x 768)
    const long t128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
    if (v == 1 && (g_gemm_variant == 0 || g_gemm_variant >= 15) && !TA && !TB && splitk <= 1 && t128 >= 160 && t128 <= 256) v = 5;
    if (v == 8 || v == 9) {      // experiments: 256x128 / 128x256 tiles, 4-wave blocks, two blocks per CU (one block's epilogue under the other's K loop)
        if (aligned && p.K % 32 == 0 && p.M >= 256 && p.N >= 256) {
            g_gemm_last_variant = v;
            if (v == 8) return launch_large<TO, TA, TB, 32, 3, 256, 128, 2, 2, 2>(p, splitk, s);
            return launch_large<TO, TA, TB, 32, 3, 128, 256, 2, 2, 2>(p, splitk, s);
        }
        v = 1;
    }
    if (v == 5) {      // 128x128 tile on the direct-to-LDS ring (4 stages): mid-size problems, one block per CU
        if (aligned && p.K % 64 == 0 && p.M >= 128 && p.N >= 128) { g_gemm_last_variant = 5; return launch_large<TO, TA, TB, 64, 4, 128, 128, 2, 2, 1>(p, splitk, s); }
        v = 1;
    }
    if (!big_ok) v = 1;
    g_gemm_last_variant = v;
    if (p.aux_blocked && !(v == 3 || (v >= 10 && v <= 13)))
        return simseg_set_error("simseg_gemm: act 5 / 6 (tile-blocked saved derivative) needs the 256x256 ping-pong kernel; this call dispatches to variant %d", v);
    if (v == 3 || (v >= 10 && v <= 13)) {
        // more than one round of full tiles with an accumulator-layout epilogue: the persistent kernel (10 forces it wherever it
        // applies, 3 forces the per-tile kernel - A/B runs)
        // (variant 10 forces it wherever it applies, 11-13 are its schedule experiments; profiles/r3_gemm_persistent_ab.txt: +6 % over the
        // per-tile kernel on the training shapes in isolation)
        if constexpr (!TA) {
            // The persistent kernel is the default for problems of >= 600 full tiles (SIMSEG_GEMM_PP2_MIN_TILES; variant 3 forces the per-tile
            // kernel).  Measured in the default two-stream training step, same box, two rounds each: 90.9-91.4 ms per-tile -> 88.1-88.4 ms
            // (thresholds 257..1100 within 0.3 ms of each other, 2000: 89.5); single stream 94.9 -> 93.8 ms.  The first half of round 3 had
            // found the opposite (100.1 vs 93.8 ms) - with variant 10, which ALSO takes the split-K weight gradients off the ping-pong kernel
            // (the `v >= 10 && !(big_ok && !TA)` rule below sends them to the 128x128 kernel): that, not the persistence, was the loss.
            // SIMSEG_GEMM_PP2_RESERVE leaves CUs to the other stream's kernels (0, 8, 32 measured within 0.4 ms; 40+ slower): default 0.
            static int reserve = -1, min_tiles = -1;
            if (reserve < 0) {
                const char* e = getenv("SIMSEG_GEMM_PP2_RESERVE"); reserve = e ? atoi(e) : 0;
                e = getenv("SIMSEG_GEMM_PP2_MIN_TILES"); min_tiles = e ? atoi(e) : 600;
            }
            // (the times-saved-derivative epilogue stays on the per-tile kernel unless SIMSEG_GEMM_PP2_ACT4=1: beside the persistent kernel's tile-walking
            //  state its 32 column-sum partials spill - 0.709 vs 0.709 ms on the fc2 dgrad, 2-6 % slower on the other shapes, tools/gemm_ab.py --act 4)
            static int act4 = -1;
            if (act4 < 0) { const char* e = getenv("SIMSEG_GEMM_PP2_ACT4"); act4 = e ? atoi(e) : 0; }
            if ((g_gemm_variant == 18 || g_gemm_variant == 0) && tiles256 >= min_tiles && (p.act != 4 || p.aux_blocked || act4) && pp2_ok<TO>(p, splitk)) {
                g_gemm_last_variant = 10;
                return launch_pp2_ek<TO, TA, TB>(p, s, reserve);
            }
            if (g_gemm_variant >= 10 && g_gemm_variant <= 13 && pp2_ok<TO>(p, splitk)) {
                g_gemm_last_variant = 10;
                if (g_gemm_variant != 10 && p.aux_blocked) return simseg_set_error("simseg_gemm: the schedule experiments (variants 11-13) do not carry the tile-blocked epilogues");
                switch (g_gemm_variant) {           // 11..13: schedule experiments (see SCHED)
                    case 11: return launch_pp2<TO, TA, TB, 1>(p, s);
                    case 12: return launch_pp2<TO, TA, TB, 2>(p, s);
                    case 13: return launch_pp2<TO, TA, TB, 3>(p, s);
                    default: return launch_pp2_ek<TO, TA, TB>(p, s, 0);
                }
            }
        }
        g_gemm_last_variant = 3;
        // 16-MFMA phases (round 3) unless variant 15 asks for the four-phase schedule (A/B runs)
        if (!four_phase && p.K >= 128) {
            if constexpr (sizeof(TO) == 2 && !TA) {
                if constexpr (!TB) { if (p.act == 3 && p.aux_blocked == 1) return launch_pp<TO, TA, TB, 4, 1, 1, 2>(p, splitk, s); }
                if constexpr (TB) { if (p.act == 4 && p.aux_blocked == 1) return launch_pp<TO, TA, TB, 4, 1, 1, 3>(p, splitk, s); }
                if constexpr (!TB) { if (p.act == 3 && p.aux_blocked == 2) return launch_pp<TO, TA, TB, 4, 1, 1, 4>(p, splitk, s); }
                if constexpr (TB) { if (p.act == 4 && p.aux_blocked == 2) return launch_pp<TO, TA, TB, 4, 1, 1, 5>(p, splitk, s); }
            }
            if (p.aux_blocked) return simseg_set_error("simseg_gemm: act 5 is a forward (x . W^T) epilogue, act 6 a dgrad (d . W) epilogue");
            if (sizeof(TO) == 2 && p.act != 4) return launch_pp<TO, TA, TB, 4, 1, 1, 0>(p, splitk, s);
            return launch_pp<TO, TA, TB, 4, 1, 1>(p, splitk, s);
        }
        if (p.aux_blocked) return simseg_set_error("simseg_gemm: the four-phase schedule (variant 15) does not carry the tile-blocked epilogues");
        return launch_pp<TO, TA, TB>(p, splitk, s);
    }
    if (v == 2) return launch_large<TO, TA, TB, 64, 2, 256, 256, 2, 4, 2>(p, splitk, s);
    return launch<bf16_t, TO, TA, TB>(p, splitk, s);
}

}  // namespace

// which kernel the calling thread's last simseg_gemm launched: 1 = 128x128 register-staged, 2 = 256x256 direct-to-LDS ring,
// 3 = 256x256 ping-pong, 4 = small-problem kernel (the measurement code labels its per-kernel timings with this instead of re-deriving the dispatch rule)
extern "C" int simseg_gemm_last_variant(void) {
    SS_HALF_FWD(simseg_gemm_last_variant); return g_gemm_last_variant; }

// debugging: the ping-pong kernel writes 5 x u64 per block (start / K loop start / K loop end / end wall-clock stamps at 100 MHz, HW_ID)
extern "C" int simseg_debug_gemm_stagger(int ticks) {
#ifndef SS_GEMM_ABLATE
    SS_CHECK(ticks < 1000, "debug_gemm_stagger: %d selects an epilogue ablation, which this build does not contain (-DSS_GEMM_ABLATE)", ticks);
#endif
    g_gemm_stagger = ticks;
    return 0;
}
extern "C" int simseg_debug_gemm_wgrad_blocks(int blocks) {
#ifndef SS_HALF
    simseg_debug_gemm_wgrad_blocks_h16(blocks);
#endif
    g_gemm_wgrad_blocks = blocks;
    return 0;
}
extern "C" int simseg_debug_gemm_trace(void* buf) { g_gemm_debug_trace = static_cast<unsigned long long*>(buf); return 0; }

extern "C" int simseg_set_gemm_variant(int v) {
#ifndef SS_HALF
    simseg_set_gemm_variant_h16(v);      // the fp16 flavour keeps its own (thread-local) selector
#endif
    g_gemm_debug_skip_epilogue = v >= 100;
    g_gemm_variant = v % 100;
    return 0;
}

namespace {
}  // namespace

extern "C" int simseg_patch_text_sim(const void* x, const void* text, float* out, int64_t M, int64_t C, int64_t K, int dtype,
                                     float eps, int normalize, void* stream) {
    SS_HALF_FWD(simseg_patch_text_sim, x, text, out, M, C, K, dtype, eps, normalize, stream);
    SS_CHECK(x && text && out, "patch_text_sim: null pointer");
    SS_CHECK(M > 0 && C > 0 && C <= 256 && K > 0 && M < (1ll << 31), "patch_text_sim: need 1 <= C <= 256 (got C=%lld)", (long long)C);
    SS_CHECK(dtype == 0 || dtype == 1, "patch_text_sim: dtype must be 0 (fp32) or 1 (bf16)");
    const int epc = dtype == 0 ? 4 : 8;
    SS_CHECK(K % epc == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)text % 16) == 0, "patch_text_sim: K must be a multiple of %d and the operands 16-byte aligned", epc);
    if (dtype == 0) return dispatch_simmap<float>(x, text, out, (int)M, (int)C, (int)K, eps, normalize, (hipStream_t)stream);
    return dispatch_simmap<bf16_t>(x, text, out, (int)M, (int)C, (int)K, eps, normalize, (hipStream_t)stream);
}

// 1 when a 16-bit M x N x K problem (k-contiguous A) runs on the 256x256 ping-pong kernels with full tiles only - the condition under which
// the fc1 forward may save GELU' as the tile-blocked accumulator image (act 5) for the dgrad through fc2 (act 6) to read back
extern "C" int simseg_gemm_aux_blocked_ok(int64_t M, int64_t N, int64_t K) {
    if (M <= 0 || N <= 0 || K <= 0 || M % 256 || N % 256 || K % 64 || K < 768) return 0;
    return (M / 256) * (N / 256) >= 96;
}

// dtype codes: 0 = fp32, 1 = bf16
extern "C" int simseg_gemm(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int64_t lda,
                           int64_t ldb, int64_t ldc, int in_dtype, int out_dtype, int transA, int transB, float alpha,
                           const float* bias, const float* rowscale, const float* residual, int64_t ldr, int act,
                           const void* aux, void* aux_out, int row_group, int res_mod, int accumulate, int splitk,
                           uint64_t drop_seed, float drop_p, float* colsum, void* stream) {
    SS_HALF_FWD(simseg_gemm, A, B, C, M, N, K, lda, ldb, ldc, in_dtype, out_dtype, transA, transB, alpha, bias, rowscale, residual, ldr, act, aux, aux_out, row_group, res_mod, accumulate, splitk, drop_seed, drop_p, colsum, stream);
    SS_CHECK(A && B && C, "simseg_gemm: null operand");
    SS_CHECK(M > 0 && N > 0 && K > 0 && M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31), "simseg_gemm: bad shape %lld x %lld x %lld",
             (long long)M, (long long)N, (long long)K);
    const int epc = in_dtype == 0 ? 4 : 8;
    SS_CHECK(in_dtype == 0 || in_dtype == 1, "simseg_gemm: in_dtype must be 0 (fp32) or 1 (bf16)");
    SS_CHECK(out_dtype == 0 || out_dtype == 1, "simseg_gemm: out_dtype must be 0 (fp32) or 1 (bf16)");
    SS_CHECK(in_dtype == 1 || out_dtype == 0, "simseg_gemm: fp32 operands produce fp32 output");
    SS_CHECK(in_dtype == 1 || (!transA && !transB), "simseg_gemm: transposed operands need bf16");
    SS_CHECK(!(transA && !transB), "simseg_gemm: (transA, !transB) is not on the path");
    // operands are fetched in 16-byte chunks along their contiguous extent; fp32 has an element-wise fallback
    const bool aligned = ((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0 && lda % epc == 0 && ldb % epc == 0 &&
                         (transA ? M : K) % epc == 0 && (transB ? N : K) % epc == 0;
    SS_CHECK(aligned || in_dtype == 0, "simseg_gemm: bf16 operands need 16-byte aligned pointers and contiguous extents / leading dimensions that are multiples of 8");
    SS_CHECK(((uintptr_t)A % 4) == 0 && ((uintptr_t)B % 4) == 0, "simseg_gemm: misaligned operand");
    SS_CHECK(splitk <= 1 || (out_dtype == 0 && act == 0 && !bias && !residual && drop_p == 0.f && !colsum),
             "simseg_gemm: split-K needs a plain fp32 accumulate epilogue");
    SS_CHECK(drop_p >= 0.f && drop_p < 1.f, "simseg_gemm: dropout p out of range");
    SS_CHECK(act >= 0 && act <= 8, "simseg_gemm: act must be 0..8");
    const int blocked = act >= 7 ? 2 : (act >= 5 ? 1 : 0);
    if (blocked) {       // 5 / 6 = 3 / 4 with the saved derivative as the tile-blocked accumulator image, 7 / 8 = the same image in 8 bits (include/simseg_hip.h)
        const bool fwd = act == 5 || act == 7;
        SS_CHECK(simseg_gemm_aux_blocked_ok(M, N, K) && in_dtype == 1 && out_dtype == 1 && !transA && !rowscale && !residual && row_group == 0 &&
                 drop_p == 0.f && splitk <= 1 && ldc == N && (fwd ? aux_out != nullptr : aux != nullptr) &&
                 ((uintptr_t)(fwd ? aux_out : aux) % 16) == 0 && ((uintptr_t)C % 16) == 0 && (!bias || ((uintptr_t)bias % 16) == 0),
                 "simseg_gemm: act %d needs a full-tile 16-bit problem on the ping-pong kernel (simseg_gemm_aux_blocked_ok) with a plain epilogue", act);
        act = fwd ? 3 : 4;
    }
    SS_CHECK((act != 2 && act != 4) || aux, "simseg_gemm: act=2/4 needs aux");
    SS_CHECK(!res_mod || row_group > 0, "simseg_gemm: res_mod needs row_group");
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.A = A; p.B = B; p.C = C; p.M = (int)M; p.N = (int)N; p.K = (int)K;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.alpha = alpha; p.bias = bias; p.rowscale = rowscale;
    p.residual = residual; p.ldr = ldr; p.act = act; p.aux = aux; p.aux_out = aux_out;
    p.row_group = row_group; p.res_mod = res_mod; p.accumulate = accumulate; p.dbg_skip_epilogue = g_gemm_debug_skip_epilogue; p.dbg_trace = g_gemm_debug_trace; p.stagger = g_gemm_stagger;
    p.drop_seed = drop_seed;
    p.colsum = colsum;
    p.drop_thresh = drop_p > 0.f ? (unsigned int)((double)drop_p * 4294967296.0) : 0u;
    p.drop_scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    p.aux_blocked = blocked;
    p.pf_next = g_gemm_variant != 16;                                // 16 = automatic choice without the next-tile fetch (A/B runs)
    hipStream_t s = (hipStream_t)stream;
    g_gemm_last_variant = 1;
    // 0 auto, 4 = the small-problem kernel wherever it applies (tests), any other value keeps it off
    const bool small_fit = !transA && !transB && aligned && splitk <= 1 && (K * (in_dtype == 0 ? 4 : 2)) % 128 == 0;
    // measured (profiles/r2_gemm_small_problem.txt): ahead of the 128x128 kernel up to ~160 of its tiles in bf16, ~200 in fp32
    const int64_t t128 = ((M + 127) / 128) * ((N + 127) / 128);
    const bool small_auto = small_fit && t128 < (in_dtype == 0 ? 200 : 160);
    if (((g_gemm_variant == 0 || g_gemm_variant >= 15) && small_auto) || (g_gemm_variant == 4 && small_fit)) {
        g_gemm_last_variant = 4;
        if (in_dtype == 0) return dispatch_small<float, float>(p, s);
        return out_dtype ? dispatch_small<bf16_t, bf16_t>(p, s) : dispatch_small<bf16_t, float>(p, s);
    }
    if (in_dtype == 0) return aligned ? launch<float, float, false, false, true>(p, splitk, s) : launch<float, float, false, false, false>(p, splitk, s);
    if (!transA && !transB) return out_dtype ? dispatch_bf16<bf16_t, false, false>(p, splitk, aligned, s) : dispatch_bf16<float, false, false>(p, splitk, aligned, s);
    if (!transA && transB) return out_dtype ? dispatch_bf16<bf16_t, false, true>(p, splitk, aligned, s) : dispatch_bf16<float, false, true>(p, splitk, aligned, s);
    return out_dtype ? dispatch_bf16<bf16_t, true, true>(p, splitk, aligned, s) : dispatch_bf16<float, true, true>(p, splitk, aligned, s);
}
