// Zero-shot segmentation post-processing on the GPU: everything the reference's evaluate_benchmark does
// per image AFTER the similarity map and BEFORE / AFTER its CPU DenseCRF (tools/seg_evaluation.py:112-170):
//   K17 seg_select   : top-`top_cls_num` class scores, threshold = mean + std, first five candidates (:112-124,:127-143)
//   K18 seg_masks    : per candidate: column of the similarity map -> min-max normalise (:145-146) -> binary (unary argmax,
//                      i.e. prob > 0.5: what dense_crf(:30-54) returns with its pairwise terms switched off) -> x16 nearest (:132)
//   K19 morph7       : 7x7 dilate / erode, one iteration each, borders ignored (cv2 defaults; :153-156)
//   K20 seg_predict  : nearest resize to the label size (:158), score-weighted argmax over classes (:159,:162), and the
//                      intersect / pred / label histograms of mean_iou (simseg/utils/metrics.py:5-75) in the same pass
// All of it is byte / index work bound by HBM traffic; nothing here touches MFMA.
#include "common.h"

namespace {

constexpr int SEL_MAXC = 2048;   // classes per image handled by the selection kernel
constexpr int SEL_MAXTOP = 64;

// K17: one block per image
__global__ __launch_bounds__(256) void seg_select_kernel(const float* __restrict__ scores, int C, int topn, int ncand,
                                                         int* __restrict__ cand_idx, float* __restrict__ cand_score,
                                                         float* __restrict__ threshold) {
    __shared__ float sc[SEL_MAXC];
    __shared__ float topv[SEL_MAXTOP];
    __shared__ int topi[SEL_MAXTOP];
    __shared__ float wv[4];
    __shared__ int wi[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int j = tid; j < C; j += 256) sc[j] = scores[(long)b * C + j];
    __syncthreads();
    for (int t = 0; t < topn; ++t) {
        float mv = -INFINITY;
        int mi = 0x7fffffff;
        for (int j = tid; j < C; j += 256) {
            const float v = sc[j];
            if (v > mv || (v == mv && j < mi)) { mv = v; mi = j; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(mv, o, 64);
            const int oi = __shfl_xor(mi, o, 64);
            if (ov > mv || (ov == mv && oi < mi)) { mv = ov; mi = oi; }
        }
        if ((tid & 63) == 0) { wv[tid >> 6] = mv; wi[tid >> 6] = mi; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 4; ++w)
                if (wv[w] > mv || (wv[w] == mv && wi[w] < mi)) { mv = wv[w]; mi = wi[w]; }
            topv[t] = mv; topi[t] = mi;
            if (mi < C) sc[mi] = -INFINITY;
        }
        __syncthreads();
    }
    if (tid == 0) {
        float mean = 0.f;
        for (int t = 0; t < topn; ++t) mean += topv[t];
        mean /= (float)topn;
        float var = 0.f;
        for (int t = 0; t < topn; ++t) var += (topv[t] - mean) * (topv[t] - mean);
        var /= (float)(topn - 1);                        // torch.std(): unbiased
        const float thr = mean + sqrtf(var);
        if (threshold) threshold[b] = thr;
        for (int i = 0; i < ncand; ++i) {
            int idx = -1;
            float s = 0.f;
            if (i < topn) {
                s = topv[i];
                // `continue` on class 0 / 255, `break` below the threshold: the scores are sorted, so both reduce to a per-slot test
                if (topi[i] != 0 && topi[i] != 255 && !(s < thr)) idx = topi[i];
            }
            cand_idx[b * ncand + i] = idx;
            cand_score[b * ncand + i] = s;
        }
    }
}

// K18: one block per (candidate, image).  sim [B, N, C] fp32; prob [B, ncand, N] (optional); mask [B, ncand, 16n, 16n] bytes.
constexpr int MASK_MAXN = 4096;
__global__ __launch_bounds__(256) void seg_mask_kernel(const float* __restrict__ sim, const int* __restrict__ cand_idx, int N, int n,
                                                       int C, int ncand, float* __restrict__ prob, unsigned char* __restrict__ mask) {
    // (n = patch columns, N / n = patch rows: square for one resized image, nh x nw for a stitched sliding-window map)
    __shared__ float v[MASK_MAXN];
    __shared__ float rmin[4], rmax[4];
    const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int idx = cand_idx[b * ncand + c];
    if (idx < 0) return;
    float mn = INFINITY, mx = -INFINITY;
    for (int i = tid; i < N; i += 256) {
        const float x = sim[((long)b * N + i) * C + idx];
        v[i] = x;
        mn = fminf(mn, x); mx = fmaxf(mx, x);
    }
    mn = wave_min(mn); mx = wave_max(mx);
    if ((tid & 63) == 0) { rmin[tid >> 6] = mn; rmax[tid >> 6] = mx; }
    __syncthreads();
    mn = fminf(fminf(rmin[0], rmin[1]), fminf(rmin[2], rmin[3]));
    mx = fmaxf(fmaxf(rmax[0], rmax[1]), fmaxf(rmax[2], rmax[3]));
    const float range = mx - mn;
    for (int i = tid; i < N; i += 256) {
        const float pr = (v[i] - mn) / range;            // a constant map gives 0/0 = NaN as in the reference: never > 0.5
        if (prob) prob[((long)b * ncand + c) * N + i] = pr;
        v[i] = pr > 0.5f ? 1.f : 0.f;
    }
    __syncthreads();
    // x16 nearest: every 16-byte store is one pixel row of one patch cell
    const int Hm = (N / n) * 16, Wm = n * 16;
    uint4* out = reinterpret_cast<uint4*>(mask + ((long)b * ncand + c) * Hm * Wm);
    const long segs = (long)Hm * n;
    for (long s = tid; s < segs; s += 256) {
        const int y = (int)(s / n), px = (int)(s % n);
        const unsigned w = v[(y >> 4) * n + px] != 0.f ? 0xffffffffu : 0u;
        out[s] = make_uint4(w, w, w, w);
    }
}

// K18b (round 5, BASELINE configs[3] "slide-window 512x512"): overlap-average of per-window maps on the source image's patch grid.
// win [B, wy, wx, n, n, C]: window (i, j) covers source patch rows i*step .. i*step+n-1 and columns j*step .. j*step+n-1 (512-pixel
// windows at stride 256 on 16-pixel patches: n = 32, step = 16); out [B, nh, nw, C] with nh = n + (wy-1)*step: the mean over the windows
// that cover each cell, summed in window order (i, then j) - the order the oracle's loop uses, so fp32 results agree bit for bit.
// step = 0 averages all windows cell by cell (n = 1: the mean of per-window class scores).  HBM-bound, read-once / write-once: one
// thread per output element, consecutive threads on consecutive classes (the contiguous axis of both tensors).
__global__ __launch_bounds__(256) void stitch_windows_kernel(const float* __restrict__ win, float* __restrict__ out, int B, int wy, int wx,
                                                             int n, int step, int C, int nh, int nw) {
    const long total = (long)B * nh * nw * C;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c = (int)(e % C);
        long r = e / C;
        const int x = (int)(r % nw); r /= nw;
        const int y = (int)(r % nh);
        const int b = (int)(r / nh);
        // windows covering row y: i*step <= y <= i*step + n - 1
        int i0 = 0, i1 = wy - 1, j0 = 0, j1 = wx - 1;
        if (step > 0) {
            i0 = max(0, (y - n + step) / step); i1 = min(wy - 1, y / step);
            j0 = max(0, (x - n + step) / step); j1 = min(wx - 1, x / step);
        }
        float acc = 0.f;
        int cnt = 0;
        for (int i = i0; i <= i1; ++i)
            for (int j = j0; j <= j1; ++j) {
                const int ly = y - i * step, lx = x - j * step;
                acc += win[((((long)b * wy + i) * wx + j) * n * n + (long)ly * n + lx) * C + c];
                ++cnt;
            }
        out[e] = acc / (float)cnt;
    }
}

// K19: 7x7 max (dilate) / min (erode) filter on byte images [M, H, W]; pixels outside the image never win.
constexpr int MT = 64, MR = 3, MTP = MT + 2 * MR;
__global__ __launch_bounds__(256) void morph7_kernel(const unsigned char* __restrict__ in, unsigned char* __restrict__ out, int H, int W,
                                                     int erode) {
    __shared__ unsigned char tile[MTP][MTP + 2];
    __shared__ unsigned char hrow[MTP][MT];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * MT, y0 = blockIdx.y * MT;
    const unsigned char* src = in + (long)blockIdx.z * H * W;
    unsigned char* dst = out + (long)blockIdx.z * H * W;
    const unsigned char neutral = erode ? 255 : 0;
    for (int i = tid; i < MTP * MTP; i += 256) {
        const int ty = i / MTP, tx = i % MTP;
        const int y = y0 + ty - MR, x = x0 + tx - MR;
        tile[ty][tx] = (y >= 0 && y < H && x >= 0 && x < W) ? src[(long)y * W + x] : neutral;
    }
    __syncthreads();
    for (int i = tid; i < MTP * MT; i += 256) {
        const int ty = i / MT, tx = i % MT;
        unsigned char m = tile[ty][tx];
#pragma unroll
        for (int d = 1; d <= 2 * MR; ++d) {
            const unsigned char t = tile[ty][tx + d];
            m = erode ? (t < m ? t : m) : (t > m ? t : m);
        }
        hrow[ty][tx] = m;
    }
    __syncthreads();
    for (int i = tid; i < MT * MT; i += 256) {
        const int ty = i / MT, tx = i % MT;
        const int y = y0 + ty, x = x0 + tx;
        if (y >= H || x >= W) continue;
        unsigned char m = hrow[ty][tx];
#pragma unroll
        for (int d = 1; d <= 2 * MR; ++d) {
            const unsigned char t = hrow[ty + d][tx];
            m = erode ? (t < m ? t : m) : (t > m ? t : m);
        }
        dst[(long)y * W + x] = m;
    }
}

// K19b: cv2.dilate followed by cv2.erode (7x7, one iteration each) in ONE pass: a 64x64 output tile needs the input with a
// halo of 6; the dilated intermediate (halo 3) lives only in LDS, and positions of it outside the image are set to 255 so
// the erosion ignores them exactly as cv2's border rule does.  `valid` (optional, one int per image) < 0 skips the image:
// slots the reference never visits (most of the five candidates of an image) cost nothing.
constexpr int CT = 64, CH = 6, CIN = CT + 2 * CH, CMID = CT + 2 * MR;
__global__ __launch_bounds__(256) void close7_kernel(const unsigned char* __restrict__ in, unsigned char* __restrict__ out, const int* __restrict__ valid,
                                                     int H, int W) {
    __shared__ unsigned char A[CIN][CIN + 4];
    __shared__ unsigned char Bh[CIN][CMID + 2];
    __shared__ unsigned char C[CMID][CMID + 2];
    __shared__ unsigned char D[CMID][CT];
    if (valid && valid[blockIdx.z] < 0) return;
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * CT, y0 = blockIdx.y * CT;
    const unsigned char* src = in + (long)blockIdx.z * H * W;
    unsigned char* dst = out + (long)blockIdx.z * H * W;
    for (int i = tid; i < CIN * CIN; i += 256) {
        const int ty = i / CIN, tx = i % CIN;
        const int y = y0 + ty - CH, x = x0 + tx - CH;
        A[ty][tx] = (y >= 0 && y < H && x >= 0 && x < W) ? src[(long)y * W + x] : 0;
    }
    __syncthreads();
    for (int i = tid; i < CIN * CMID; i += 256) {            // horizontal max: Bh[y][x] covers input columns x .. x+6
        const int ty = i / CMID, tx = i % CMID;
        unsigned char m = A[ty][tx];
#pragma unroll
        for (int d = 1; d <= 2 * MR; ++d) { const unsigned char t = A[ty][tx + d]; m = t > m ? t : m; }
        Bh[ty][tx] = m;
    }
    __syncthreads();
    for (int i = tid; i < CMID * CMID; i += 256) {           // vertical max -> dilated value at (y0 - 3 + ty, x0 - 3 + tx)
        const int ty = i / CMID, tx = i % CMID;
        unsigned char m = Bh[ty][tx];
#pragma unroll
        for (int d = 1; d <= 2 * MR; ++d) { const unsigned char t = Bh[ty + d][tx]; m = t > m ? t : m; }
        const int y = y0 - MR + ty, x = x0 - MR + tx;
        C[ty][tx] = (y >= 0 && y < H && x >= 0 && x < W) ? m : 255;
    }
    __syncthreads();
    for (int i = tid; i < CMID * CT; i += 256) {             // horizontal min
        const int ty = i / CT, tx = i % CT;
        unsigned char m = C[ty][tx];
#pragma unroll
        for (int d = 1; d <= 2 * MR; ++d) { const unsigned char t = C[ty][tx + d]; m = t < m ? t : m; }
        D[ty][tx] = m;
    }
    __syncthreads();
    for (int i = tid; i < CT * CT; i += 256) {               // vertical min -> output
        const int ty = i / CT, tx = i % CT;
        const int y = y0 + ty, x = x0 + tx;
        if (y >= H || x >= W) continue;
        unsigned char m = D[ty][tx];
#pragma unroll
        for (int d = 1; d <= 2 * MR; ++d) { const unsigned char t = D[ty + d][tx]; m = t < m ? t : m; }
        dst[(long)y * W + x] = m;
    }
}

// K20: masks [B, ncand, Hm, Wm] -> pred [B, H, W] (optional) + hist [3, C] += (intersect, pred, label) pixel counts.
constexpr int HIST_MAXC = 1024;
__global__ __launch_bounds__(256) void seg_predict_kernel(const unsigned char* __restrict__ masks, const int* __restrict__ cand_idx,
                                                          const float* __restrict__ cand_score, const unsigned char* __restrict__ labels,
                                                          int ncand, int Hm, int Wm, int H, int W, int C, int ignore,
                                                          int* __restrict__ pred, unsigned long long* __restrict__ hist) {
    __shared__ unsigned int lh[3 * HIST_MAXC];
    __shared__ int cidx[8];
    __shared__ double cval[8];
    const int b = blockIdx.y, tid = threadIdx.x;
    for (int i = tid; i < 3 * C; i += 256) lh[i] = 0u;
    if (tid < ncand) {
        cidx[tid] = cand_idx[b * ncand + tid];
        cval[tid] = (double)cand_score[b * ncand + tid];
    }
    __syncthreads();
    const double fy = (double)Hm / (double)H, fx = (double)Wm / (double)W;      // cv2.resize INTER_NEAREST: src = floor(dst * scale)
    const long npix = (long)H * W;
    if (Hm == H && Wm == W && (npix & 3) == 0 && !pred) {
        // same-size maps (windowed evaluation), histograms only: four pixels per thread, one 32-bit load per candidate map
        const unsigned int* lab4 = reinterpret_cast<const unsigned int*>(labels + (long)b * npix);
        for (long i = (long)blockIdx.x * 256 + tid; i < (npix >> 2); i += (long)gridDim.x * 256) {
            unsigned int mk[8];
            for (int k = 0; k < ncand; ++k)
                mk[k] = cidx[k] >= 0 ? reinterpret_cast<const unsigned int*>(masks + ((long)b * ncand + k) * npix)[i] : 0u;
            const unsigned int lw = lab4[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                double best = 0.0;
                int bi = 0;
                for (int k = 0; k < ncand; ++k) {
                    const int ci = cidx[k];
                    if (ci < 0) continue;
                    const double val = (double)((mk[k] >> (8 * e)) & 0xffu) * cval[k];
                    if (val > best || (val == best && ci < bi)) { best = val; bi = ci; }
                }
                const int l = (int)((lw >> (8 * e)) & 0xffu);
                // label maps are spatially coherent: most waves see ONE (prediction, label) pair, which would be a 64-way
                // conflict on three LDS counters.  Lanes agreeing with the first lane are counted by one lane.
                const int key = (bi << 8) | l;
                const int k0 = __builtin_amdgcn_readfirstlane(key);
                const unsigned long long same = __ballot(key == k0);
                const bool leader = (key == k0) && (__builtin_amdgcn_mbcnt_hi((unsigned)(same >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)same, 0u)) == 0);
                const unsigned cnt = (key == k0) ? (leader ? (unsigned)__popcll(same) : 0u) : 1u;
                if (cnt && l != ignore) {
                    atomicAdd(&lh[C + bi], cnt);
                    if (l < C) {
                        atomicAdd(&lh[2 * C + l], cnt);
                        if (l == bi) atomicAdd(&lh[l], cnt);
                    }
                }
            }
        }
        __syncthreads();
        for (int i = tid; i < 3 * C; i += 256)
            if (lh[i]) atomicAdd(&hist[i], (unsigned long long)lh[i]);
        return;
    }
    for (long i = (long)blockIdx.x * 256 + tid; i < npix; i += (long)gridDim.x * 256) {
        const int y = (int)(i / W), x = (int)(i % W);
        int sy = (int)floor(y * fy), sx = (int)floor(x * fx);
        sy = sy < Hm - 1 ? sy : Hm - 1; sx = sx < Wm - 1 ? sx : Wm - 1;
        // temp_pred[class] = mask * score (float64), argmax over classes = first class attaining the maximum; class 0 is
        // never a candidate and holds 0, so only strictly positive values can beat it
        double best = 0.0;
        int bi = 0;
        for (int k = 0; k < ncand; ++k) {
            const int ci = cidx[k];
            if (ci < 0) continue;
            const unsigned char m = masks[(((long)b * ncand + k) * Hm + sy) * Wm + sx];
            const double val = (double)m * cval[k];
            if (val > best || (val == best && ci < bi)) { best = val; bi = ci; }
        }
        if (pred) pred[(long)b * npix + i] = bi;
        const int l = labels[(long)b * npix + i];
        if (l != ignore) {                                                       // metrics.py:60-66
            atomicAdd(&lh[C + bi], 1u);
            if (l < C) {
                atomicAdd(&lh[2 * C + l], 1u);
                if (l == bi) atomicAdd(&lh[l], 1u);
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < 3 * C; i += 256)
        if (lh[i]) atomicAdd(&hist[i], (unsigned long long)lh[i]);
}

}  // namespace

extern "C" int simseg_seg_select(const float* scores, int* cand_idx, float* cand_score, float* threshold, int64_t B, int64_t C,
                                 int64_t top_cls_num, int64_t ncand, void* stream) {
    SS_CHECK(scores && cand_idx && cand_score, "seg_select: null pointer");
    SS_CHECK(B > 0 && C > 1 && C <= SEL_MAXC, "seg_select: need 1 < C <= %d", SEL_MAXC);
    SS_CHECK(ncand >= 1 && ncand <= 8, "seg_select: 1 <= ncand <= 8");
    const int topn = (int)(top_cls_num < C ? top_cls_num : C);
    SS_CHECK(topn >= 2 && topn <= SEL_MAXTOP, "seg_select: 2 <= min(top_cls_num, C) <= %d", SEL_MAXTOP);
    hipLaunchKernelGGL(seg_select_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, scores, (int)C, topn, (int)ncand, cand_idx,
                       cand_score, threshold);
    SS_LAUNCH_CHECK("seg_select");
    return 0;
}

extern "C" int simseg_seg_masks_rect(const float* sim, const int* cand_idx, float* prob, void* mask, int64_t B, int64_t nh, int64_t nw,
                                     int64_t C, int64_t ncand, void* stream) {
    SS_CHECK(sim && cand_idx && mask, "seg_masks: null pointer");
    SS_CHECK(B > 0 && nh > 0 && nw > 0 && nh * nw <= MASK_MAXN && C > 0 && ncand >= 1 && ncand <= 8, "seg_masks: bad shape (nh*nw <= %d)", MASK_MAXN);
    SS_CHECK(((uintptr_t)mask % 16) == 0, "seg_masks: mask must be 16-byte aligned");
    hipLaunchKernelGGL(seg_mask_kernel, dim3((unsigned)ncand, (unsigned)B), dim3(256), 0, (hipStream_t)stream, sim, cand_idx, (int)(nh * nw),
                       (int)nw, (int)C, (int)ncand, prob, static_cast<unsigned char*>(mask));
    SS_LAUNCH_CHECK("seg_masks");
    return 0;
}

extern "C" int simseg_seg_masks(const float* sim, const int* cand_idx, float* prob, void* mask, int64_t B, int64_t n, int64_t C,
                                int64_t ncand, void* stream) {
    return simseg_seg_masks_rect(sim, cand_idx, prob, mask, B, n, n, C, ncand, stream);
}

extern "C" int simseg_stitch_windows(const float* win, float* out, int64_t B, int64_t wy, int64_t wx, int64_t n, int64_t step, int64_t C,
                                     void* stream) {
    SS_CHECK(win && out && win != out, "stitch_windows: null or aliased pointers");
    SS_CHECK(B > 0 && wy > 0 && wx > 0 && n > 0 && step >= 0 && step <= n && C > 0, "stitch_windows: bad shape (0 <= step <= n)");
    const long nh = n + (wy - 1) * step, nw = n + (wx - 1) * step;
    const long total = (long)B * nh * nw * C;
    SS_CHECK(total < (1ll << 40), "stitch_windows: problem too large");
    long blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(stitch_windows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, win, out, (int)B, (int)wy, (int)wx, (int)n,
                       (int)step, (int)C, (int)nh, (int)nw);
    SS_LAUNCH_CHECK("stitch_windows");
    return 0;
}

extern "C" int simseg_morph7(const void* in, void* out, int64_t M, int64_t H, int64_t W, int erode, void* stream) {
    SS_CHECK(in && out && in != out, "morph7: null or aliased pointers");
    SS_CHECK(M > 0 && M < 65536 && H > 0 && W > 0, "morph7: bad shape");
    dim3 grid((unsigned)((W + MT - 1) / MT), (unsigned)((H + MT - 1) / MT), (unsigned)M);
    hipLaunchKernelGGL(morph7_kernel, grid, dim3(256), 0, (hipStream_t)stream, static_cast<const unsigned char*>(in),
                       static_cast<unsigned char*>(out), (int)H, (int)W, erode);
    SS_LAUNCH_CHECK("morph7");
    return 0;
}

extern "C" int simseg_close7(const void* in, void* out, const int* valid, int64_t M, int64_t H, int64_t W, void* stream) {
    SS_CHECK(in && out && in != out, "close7: null or aliased pointers");
    SS_CHECK(M > 0 && M < 65536 && H > 0 && W > 0, "close7: bad shape");
    dim3 grid((unsigned)((W + CT - 1) / CT), (unsigned)((H + CT - 1) / CT), (unsigned)M);
    hipLaunchKernelGGL(close7_kernel, grid, dim3(256), 0, (hipStream_t)stream, static_cast<const unsigned char*>(in),
                       static_cast<unsigned char*>(out), valid, (int)H, (int)W);
    SS_LAUNCH_CHECK("close7");
    return 0;
}

extern "C" int simseg_seg_predict(const void* masks, const int* cand_idx, const float* cand_score, const void* labels, int* pred,
                                  void* hist, int64_t B, int64_t ncand, int64_t Hm, int64_t Wm, int64_t H, int64_t W, int64_t C,
                                  int64_t ignore_index, void* stream) {
    SS_CHECK(masks && cand_idx && cand_score && labels && hist, "seg_predict: null pointer");
    SS_CHECK(B > 0 && B < 65536 && ncand >= 1 && ncand <= 8 && Hm > 0 && Wm > 0 && H > 0 && W > 0, "seg_predict: bad shape");
    SS_CHECK(C > 0 && C <= HIST_MAXC, "seg_predict: C <= %d", HIST_MAXC);
    const long npix = (long)H * W;
    long blocks = (npix + 256 * 8 - 1) / (256 * 8);          // ~8 pixels per thread: the LDS histogram flush stays small
    if (blocks < 1) blocks = 1;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(seg_predict_kernel, dim3((unsigned)blocks, (unsigned)B), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const unsigned char*>(masks), cand_idx, cand_score, static_cast<const unsigned char*>(labels), (int)ncand,
                       (int)Hm, (int)Wm, (int)H, (int)W, (int)C, (int)ignore_index, pred, static_cast<unsigned long long*>(hist));
    SS_LAUNCH_CHECK("seg_predict");
    return 0;
}
