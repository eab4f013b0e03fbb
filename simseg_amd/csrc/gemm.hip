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

namespace {

constexpr int BM = 128, BN = 128, NTHREADS = 256;
constexpr int KC_PITCH = 144;              // k-contiguous tile: 128 rows x (128 B + 16 B pad)
constexpr int MC_PITCH = 320;              // transposed tile (bf16): 64 k-rows x (256 B + 64 B pad)
constexpr int TILE_BYTES = 64 * MC_PITCH;  // 20480 >= 128 * 144

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
    int ksplit;              // k-tiles per z-slice
    // dropout on (acc*alpha + bias), before the residual:  keep iff hash(seed, row*N+col) >= thresh
    unsigned long long drop_seed; unsigned int drop_thresh; float drop_scale;
};

template <typename T> struct TT;
template <> struct TT<float> { static constexpr int EPC = 4, BK = 32; };
template <> struct TT<bf16_t> { static constexpr int EPC = 8, BK = 64; };

// ---- global -> register staging ----------------------------------------------------------------
template <typename T, bool TRANS, bool AL = true>
__device__ __forceinline__ void load_tile(u32x4 (&r)[4], const T* __restrict__ base, long ld, int row0, int lim,
                                          int k0, int K, int tid) {
    constexpr int EPC = TT<T>::EPC;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + NTHREADS * i;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (!TRANS) {
            const int rr = row0 + (idx >> 3), kk = k0 + (idx & 7) * EPC;
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

template <bool TRANS>
__device__ __forceinline__ void store_tile(const u32x4 (&r)[4], char* lds, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + NTHREADS * i;
        const int off = TRANS ? (idx >> 4) * MC_PITCH + (idx & 15) * 16 : (idx >> 3) * KC_PITCH + (idx & 7) * 16;
        *reinterpret_cast<u32x4*>(lds + off) = r[i];
    }
}

// ---- LDS -> MFMA fragment ---------------------------------------------------------------------
// Returns the 16 bytes of k this lane feeds to the MFMA for tile row (row32 + lane%32), k-step kk.
template <bool TRANS>
__device__ __forceinline__ u32x4 read_frag(const char* lds, int row32, int kk, int lane) {
    if (!TRANS) {
        return *reinterpret_cast<const u32x4*>(lds + (row32 + (lane & 31)) * KC_PITCH + (2 * kk + (lane >> 5)) * 16);
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
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua.h, ub.h, acc, 0, 0, 0);
    } else {
        union { u32x4 v; float f[4]; } ua, ub;
        ua.v = a; ub.v = b;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ua.f[j], ub.f[j], acc, 0, 0, 0);
    }
}

template <typename TO> __device__ __forceinline__ float ld_out(const TO* p) { return (float)*p; }
template <typename TO> __device__ __forceinline__ void st_out(TO* p, float v) { *p = (TO)v; }

template <typename T, typename TO, bool TA, bool TB, bool AL = true>
__global__ __launch_bounds__(NTHREADS) void gemm_kernel(GemmParams p) {
    __shared__ __attribute__((aligned(16))) char lds[2 * TILE_BYTES];
    char* ldsA = lds;
    char* ldsB = lds + TILE_BYTES;
    constexpr int BK = TT<T>::BK;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int nblk = gridDim.x;
    const int t = xcd_remap(blockIdx.x, nblk);
    const int m0 = (t / tiles_n) * BM, n0 = (t % tiles_n) * BN;

    const int nk = (p.K + BK - 1) / BK;
    const int kt0 = blockIdx.z * p.ksplit;
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

    u32x4 ra[4], rb[4];
    if (kt0 < kt1) {
        load_tile<T, TA, AL>(ra, A, p.lda, m0, p.M, kt0 * BK, p.K, tid);
        load_tile<T, TB, AL>(rb, B, p.ldb, n0, p.N, kt0 * BK, p.K, tid);
    }
    for (int kt = kt0; kt < kt1; ++kt) {
        store_tile<TA>(ra, ldsA, tid);
        store_tile<TB>(rb, ldsB, tid);
        __syncthreads();
        if (kt + 1 < kt1) {
            load_tile<T, TA, AL>(ra, A, p.lda, m0, p.M, (kt + 1) * BK, p.K, tid);
            load_tile<T, TB, AL>(rb, B, p.ldb, n0, p.N, (kt + 1) * BK, p.K, tid);
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            u32x4 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = read_frag<TA>(ldsA, wm * 64 + i * 32, kk, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = read_frag<TB>(ldsB, wn * 64 + j * 32, kk, lane);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mma<T>(acc[i][j], fa[i], fb[j]);
        }
        __syncthreads();
    }

    // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane%32, row = (r%4) + 8*(r/4) + 4*(lane/32)
    TO* C = static_cast<TO*>(p.C);
    const TO* aux = static_cast<const TO*>(p.aux);
    TO* aux_out = static_cast<TO*>(p.aux_out);
    const bool atomic = gridDim.z > 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + (lane & 31);
            if (col >= p.N) continue;
            const float bias = p.bias ? p.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row >= p.M) continue;
                float v = acc[i][j][r] * p.alpha;
                if (p.rowscale) v *= p.rowscale[row];
                v += bias;
                long orow = row;
                if (p.row_group > 0) orow = (long)(row / p.row_group) * (p.row_group + 1) + 1 + row % p.row_group;
                const long o = orow * p.ldc + col;
                if (p.act == 1) {
                    if (aux_out) st_out(aux_out + o, v);
                    v = gelu_erf(v);
                } else if (p.act == 2) {
                    v *= dgelu_erf(ld_out(aux + o));
                }
                if (p.drop_thresh) {
                    v = dropout_keep(p.drop_seed, (unsigned long long)row * p.N + col, p.drop_thresh) ? v * p.drop_scale : 0.f;
                }
                if (p.residual) {
                    const long rrow = p.res_mod ? 1 + row % p.row_group : orow;
                    v += p.residual[rrow * p.ldr + col];
                }
                if constexpr (sizeof(TO) == 4) {
                    if (atomic) { atomicAdd(reinterpret_cast<float*>(C) + o, v); continue; }
                    if (p.accumulate) v += ld_out(C + o);
                }
                st_out(C + o, v);
            }
        }
    }
}

template <typename T, typename TO, bool TA, bool TB, bool AL = true>
int launch(const GemmParams& p, int splitk, hipStream_t stream) {
    constexpr int BK = TT<T>::BK;
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    const int nk = (p.K + BK - 1) / BK;
    GemmParams q = p;
    if (splitk < 1) splitk = 1;
    if (splitk > nk) splitk = nk > 0 ? nk : 1;
    q.ksplit = (nk + splitk - 1) / splitk;
    if (q.ksplit < 1) q.ksplit = 1;
    const int z = nk > 0 ? (nk + q.ksplit - 1) / q.ksplit : 1;
    dim3 grid(tiles, 1, z);
    hipLaunchKernelGGL((gemm_kernel<T, TO, TA, TB, AL>), grid, dim3(NTHREADS), 0, stream, q);
    SS_LAUNCH_CHECK("simseg_gemm");
    return 0;
}

}  // namespace

// dtype codes: 0 = fp32, 1 = bf16
extern "C" int simseg_gemm(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int64_t lda,
                           int64_t ldb, int64_t ldc, int in_dtype, int out_dtype, int transA, int transB, float alpha,
                           const float* bias, const float* rowscale, const float* residual, int64_t ldr, int act,
                           const void* aux, void* aux_out, int row_group, int res_mod, int accumulate, int splitk,
                           uint64_t drop_seed, float drop_p, void* stream) {
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
    SS_CHECK(splitk <= 1 || (out_dtype == 0 && act == 0 && !bias && !residual && drop_p == 0.f),
             "simseg_gemm: split-K needs a plain fp32 accumulate epilogue");
    SS_CHECK(drop_p >= 0.f && drop_p < 1.f, "simseg_gemm: dropout p out of range");
    SS_CHECK(act != 2 || aux, "simseg_gemm: act=2 needs aux");
    SS_CHECK(!res_mod || row_group > 0, "simseg_gemm: res_mod needs row_group");
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.A = A; p.B = B; p.C = C; p.M = (int)M; p.N = (int)N; p.K = (int)K;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.alpha = alpha; p.bias = bias; p.rowscale = rowscale;
    p.residual = residual; p.ldr = ldr; p.act = act; p.aux = aux; p.aux_out = aux_out;
    p.row_group = row_group; p.res_mod = res_mod; p.accumulate = accumulate;
    p.drop_seed = drop_seed;
    p.drop_thresh = drop_p > 0.f ? (unsigned int)((double)drop_p * 4294967296.0) : 0u;
    p.drop_scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    hipStream_t s = (hipStream_t)stream;
    if (in_dtype == 0) return aligned ? launch<float, float, false, false, true>(p, splitk, s) : launch<float, float, false, false, false>(p, splitk, s);
    if (!transA && !transB) return out_dtype ? launch<bf16_t, bf16_t, false, false>(p, splitk, s) : launch<bf16_t, float, false, false>(p, splitk, s);
    if (!transA && transB) return out_dtype ? launch<bf16_t, bf16_t, false, true>(p, splitk, s) : launch<bf16_t, float, false, true>(p, splitk, s);
    return out_dtype ? launch<bf16_t, bf16_t, true, true>(p, splitk, s) : launch<bf16_t, float, true, true>(p, splitk, s);
}
