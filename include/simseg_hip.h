/* libsimseg_hip.so -- C ABI of the MI355X-native SimSeg hot path (gfx950 only).
 *
 * The reference (muyangyi/SimSeg) has no native boundary: its hot path is torch/timm/HF Python.  This header is
 * the boundary our Python mirror of `simseg.models` binds through ctypes; each entry point names the reference
 * code it replaces (paths relative to /root/reference, or the third-party module the reference calls).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into caller-owned memory (PyTorch-ROCm tensors); the library never
 *     allocates, frees or retains device memory;
 *   - all work is enqueued on the caller's `stream` (a hipStream_t passed as void*); nothing synchronises;
 *   - return 0 on success, negative on error; the message is in simseg_last_error() (thread-local);
 *   - dtype codes: 0 = fp32, 1 = bf16.  Matrices are row-major.
 *   - the compute entry points are stateless and re-entrant: everything a call needs is in its arguments.  The only library state
 *     is per THREAD (thread_local): the last error string and the test / benchmark selectors `simseg_set_gemm_variant`,
 *     `simseg_set_attention_variant` (default 0 = auto), which affect only later calls made by the thread that set them.
 */
#ifndef SIMSEG_HIP_H
#define SIMSEG_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int simseg_version(void);
/* Which 16-bit type the calling thread's next calls compute in: 1 = bf16 (default), 2 = IEEE fp16 - the type of the reference's AMP mode
 * (torch.cuda.amp.autocast() + GradScaler, simseg/tasks/clip/clip_runner.py:226-230, core/hooks/optimizer.py:73-82).  dtype code 1
 * ("bf16") in every signature below means "the selected 16-bit type"; both flavours are the same kernels (v_mfma_f32_32x32x16_bf16 /
 * _f16, fp32 accumulation).  Thread-local. */
int simseg_set_half_type(int t);
const char* simseg_last_error(void);

/* C[M,N] = epilogue(alpha * opA(A) . opB(B)).  transA=0: A is [M,K]; 1: [K,M].  transB=0: B is [N,K] (nn.Linear
 * weight layout); 1: [K,N].  Epilogue order: *rowscale[row], +bias[col], (save pre-activation to aux_out),
 * act (1 = erf-GELU, 2 = multiply by GELU'(aux); 3 = erf-GELU with aux_out receiving GELU'(pre-activation) instead of the
 * pre-activation, 4 = multiply by aux: the training pair, no transcendental work in backward), dropout(p, seed), +residual.  row_group=G>0 writes row r to
 * (r/G)*(G+1)+1+r%G (ViT patch rows behind [cls]); res_mod reads the residual at row 1+r%G (pos_embed).
 * splitk>1 accumulates fp32 partials atomically into C (C must hold the value to accumulate onto).
 * colsum (optional, [N]) += column sums of the stored output: the bias gradient of the layer that produced C's input.
 * Replaces: nn.Linear inside timm Block / HF BertLayer (called via simseg/models/backbones/mml/vit_builder.py:18,
 * huggingface_builder.py:16-17), Conv2d patch embed (vit_builder.py:14), SimpleProjection
 * (simseg/models/components/projection.py:45-46), the logits matmul (simseg/models/criteria/losses/mml_loss.py:73),
 * the patch x text contraction (tools/seg_evaluation.py:136) and emb_sim (simseg/tasks/clip/hooks/utils.py:36). */
int simseg_gemm(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
                int64_t ldc, int in_dtype, int out_dtype, int transA, int transB, float alpha, const float* bias,
                const float* rowscale, const float* residual, int64_t ldr, int act, const void* aux, void* aux_out,
                int row_group, int res_mod, int accumulate, int splitk, uint64_t drop_seed, float drop_p, float* colsum,
                void* stream);
/* act codes 5 / 6 of simseg_gemm = 3 / 4 (GELU with its derivative saved / times the saved derivative + column sums) with the saved tensor in
 * a tile-blocked layout private to the two calls: the fc1 forward and the dgrad through fc2 are the same M x N x K problem on the same
 * 256x256 tiling, so the derivative is stored as it lies in the accumulator registers and read back the same way (no transposition on
 * either side).  Allowed when this returns 1 for the problem (full tiles on the ping-pong kernels); aux / aux_out stay [M, N] 16-bit
 * allocations whose content is opaque.
 * act codes 7 / 8 (round 4) = 5 / 6 with that image in ONE BYTE per element (aux / aux_out: M x N bytes): GELU' lies in [-0.129, 1.129] and is
 * stored as round((g + 0.132) * 255 / 1.264), a uniform grid of 0.005 - finer than a 16-bit float's spacing where most of the gradient's
 * energy is (g in [0.5, 1.13]) and coarser where |g| is small; the product dY * g it feeds is as accurate against the exact derivative as
 * with the 16-bit image (relative RMS error 2.7e-3 against 2.5e-3, the product's own 16-bit rounding being 1.7e-3; tests/test_gpu_kernels.py)
 * at half of the largest epilogue stream of a training step. */
int simseg_gemm_aux_blocked_ok(int64_t M, int64_t N, int64_t K);

/* K14, the dense zero-shot segmentation map: out[m,c] = < x[m,:] / max(||x[m,:]||, eps), text[c,:] > for every patch row m
 * and every class c (C <= 256) in one pass over x, the row L2-normalisation (F.normalize, tools/seg_evaluation.py:112) fused
 * into the MFMA kernel; normalize=0 gives the plain x . text^T.  Replaces the per-class GEMV loop of
 * tools/seg_evaluation.py:128-139.  x[M,K], text[C,K] in `dtype` (0 fp32 exact, 1 bf16), out fp32 [M,C]. */
int simseg_patch_text_sim(const void* x, const void* text, float* out, int64_t M, int64_t C, int64_t K, int dtype, float eps,
                          int normalize, void* stream);

/* Kernel selection for benchmarking / tests (thread-local; production callers never call it): 0 auto, 1 = 128x128 register-staged kernel, 2 = 256x256 direct-to-LDS ring kernel,
 * 3 = 256x256 ping-pong kernel, 4 = small-problem kernel (32x64 / 64x64 tiles, intra-block split-K) wherever it applies (+100: debug, epilogue skipped and a cycle-counter timeline written to C by
 * tools/dbg_gemm_timeline.py). */
int simseg_set_gemm_variant(int v);
/* Which of those kernels (1 / 2 / 3 / 4) the calling thread's last simseg_gemm launched. */
int simseg_gemm_last_variant(void);

/* LayerNorm over the last dim of x[rows,D] (fp32 residual stream) -> y (out_dtype) and optionally a bf16 copy.
 * Saves mean/rstd when non-null.  Replaces nn.LayerNorm in timm Block.norm1/norm2/VisionTransformer.norm
 * (eps 1e-6) and HF Bert*Output.LayerNorm / BertEmbeddings.LayerNorm (eps 1e-12). */
int simseg_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y, int out_dtype, void* y_bf16,
                         float* mean, float* rstd, int64_t rows, int64_t D, float eps, void* stream);
/* dx = LN'(dy_bf16 + dy_f32) + dres + dres_bf16; writes dx_f32 and/or dx_bf16 (the bf16 copy optionally with the dropout mask
 * (drop_seed, drop_p) of the producing dense layer re-applied); dgamma/dbeta and dxsum (column sums of dx_bf16's values,
 * optional) are ACCUMULATED.  dres_bf16 (round 4, optional): the residual-stream gradient as the 16-bit copy the previous
 * call wrote - in the 16-bit training modes the ViT blocks hand it from LayerNorm backward to LayerNorm backward in that form
 * and no fp32 image of it is written (dx_f32 = NULL): every GEMM that consumes it reads 16-bit operands anyway.
 * y_bf16 + beta (round 4, optional): the layer's saved 16-bit OUTPUT y = xhat * gamma + beta; the normalised value xhat is then taken from it
 * ((y - beta) / gamma) instead of from the fp32 input x for every 4-channel chunk with |gamma| >= 0.05 and |beta| <= 4 |gamma| (half the bytes
 * of the kernel's largest read); the other chunks - and everything when y_bf16 is NULL - read x, which must always be given. */
int simseg_layernorm_bwd(const void* dy_bf16, const float* dy_f32, const float* dres, const void* dres_bf16, const float* x,
                         const void* y_bf16, const float* beta, const float* mean,
                         const float* rstd, const float* gamma, float* dx_f32, void* dx_bf16, float* dgamma,
                         float* dbeta, float* dxsum, float* partials, int64_t rows, int64_t D, uint64_t drop_seed, float drop_p,
                         void* stream);
/* Bytes of the `partials` workspace of simseg_layernorm_bwd: per-block column sums, folded by a second small kernel
 * (one add per column) instead of one atomic per column per block.  partials = NULL selects the atomic path. */
int64_t simseg_layernorm_bwd_workspace_bytes(int64_t rows, int64_t D);

/* out[n] += sum_r in[r,n]  (bias / embedding-table gradients). */
int simseg_colsum_accum(const void* in, int in_dtype, float* out, int64_t rows, int64_t N, int64_t ld, void* stream);

/* timm PatchEmbed (Conv2d 3->D, k16 s16) as a GEMM: gathers image[B,3,H,W] into cols[B*N,768] in (c,kh,kw) order. */
int simseg_vit_im2col(const float* image, void* cols, int out_dtype, int64_t B, int64_t H, int64_t W, void* stream);
/* x[b,0,:] = cls + pos[0] (vit_builder.py:15-17) and its gradient dcls += sum_b dx[b,0,:]. */
int simseg_vit_cls_rows(const float* cls, const float* pos, float* x, int64_t B, int64_t T, int64_t D, void* stream);
int simseg_vit_cls_grad(const float* dx, float* dcls, int64_t B, int64_t T, int64_t D, void* stream);

/* HF BertEmbeddings: out[b,l,:] = word[ids[b,l]] + pos[l] + type[0] (LayerNorm is a separate call).
 * bwd: dword[ids] += dsum for unmasked tokens (masked tokens carry an exactly-zero gradient). */
int simseg_bert_embed_fwd(const int64_t* ids, const float* word, const float* pos, const float* type0, float* out,
                          int64_t B, int64_t L, int64_t D, int64_t vocab, void* stream);
int simseg_bert_embed_bwd(const int64_t* ids, const int64_t* mask, const float* dsum, float* dword, int64_t B, int64_t L,
                          int64_t D, int64_t vocab, void* stream);

/* LoDA pooling + L2norm: emb[b,:] = l2norm(mean over the k largest tokens per channel) with the reference's
 * -10000 overwrite of masked tokens.  simseg/models/components/pooling.py:52-65 + normalization.py:6-11
 * (pipelines/clip.py:87-93,111-120).  idx[B,k,P] and norm[B] are saved for the backward.  normalize=0 returns the
 * pooled vector itself (TopKPooling used on its own). */
int simseg_topk_pool_l2norm_fwd(const void* tok, int dtype, const int64_t* mask, float* emb, int32_t* idx, float* norm,
                                float* scratch, int64_t B, int64_t N, int64_t P, int k, float eps, int normalize, void* stream);
/* bytes of the optional `scratch` workspace: with it the tokens of an image are scanned by 32 blocks in parallel and merged (same
 * result, ties included) - for small batches, where one block per image leaves the GPU empty; NULL = one block per image. */
int64_t simseg_topk_pool_workspace_bytes(int64_t B, int64_t P, int k);
int simseg_topk_pool_l2norm_bwd(const float* demb, const float* emb, const float* norm, const int32_t* idx, void* dtok,
                                int dtype, int64_t B, int64_t N, int64_t P, int k, float eps, int normalize, void* stream);

/* Kernel selection for benchmarking / tests (thread-local): 0 auto (bf16 sequences of <= 256 tokens run the "resident" kernels that hold a
 * whole head's K / V - or Q / dO - in LDS), 1 = always the streaming ring kernels (2 / 3: timing ablations). */
int simseg_set_attention_variant(int v);
int simseg_debug_attn_occupancy(int64_t T);
/* debug (thread-local): the resident dK/dV kernel writes 3 x uint64 per block (start / operands landed / end, 100 MHz wall clock). */
int simseg_debug_attn_trace(void* buf);
/* debug (thread-local): the ping-pong GEMM kernel writes 5 x uint64 per block into buf (wall-clock stamps at 100 MHz of block start, K loop
 * start, K loop end, block end; HW_ID) - tools/dbg_gemm_trace.py; NULL switches it off. */
int simseg_debug_gemm_trace(void* buf);
int simseg_debug_gemm_stagger(int ticks);
/* debug / experiments (thread-local): > 0 = the split-K weight-gradient GEMMs (transA, fp32 accumulate) use at most this many 256x256
 * blocks, i.e. CUs, instead of one round of all of them; 0 = default. */
int simseg_debug_gemm_wgrad_blocks(int blocks);

/* Fused softmax attention, head_dim 64, from the packed projection qkv[B,T,3,H,64] to ctx[B,T,H*64].
 * key_mask[B,T] (1 = attend, 0 = padding; may be NULL) reproduces HF's additive key-padding mask; lse[B,H,T]
 * (log2 domain, optional) is saved for the backward.  dtype 0 = exact fp32 MFMA, 1 = bf16 MFMA.  drop_p > 0 applies
 * HF attention_probs dropout.  Replaces timm Attention.forward (q@k^T*scale, softmax, @v) and HF
 * BertSelfAttention.forward as reached through vit_builder.py:18 / huggingface_builder.py:16-17. */
/* skip_padded_rows (bf16, T <= 256, with a key mask): the caller neither reads the outputs nor uses the gradients of the rows past a
 * sequence's last unmasked key (the packed text tower): the kernels then work on 1 + that key's index rows of each sequence only and
 * leave the other rows of out / lse / dqkv untouched. */
int simseg_attention_fwd(const void* qkv, const int64_t* key_mask, void* out, float* lse, int dtype, int64_t B, int64_t T,
                         int64_t H, float scale, uint64_t drop_seed, float drop_p, int skip_padded_rows, void* stream);
/* The same forward for long 16-bit sequences (T >= 512, no mask, no dropout: timm Attention.forward at img_size 384 / 512,
 * vit_builder.py:18 via pipelines/clip.py:193-194) on a projection whose q columns already carry scale * log2(e): the caller folds the
 * factor into the q rows of Attention.qkv's weight and bias BEFORE rounding them to 16 bits, so q * scale is rounded once - exactly as q
 * alone would have been - and the kernel's exponent is the MFMA output itself (no multiply per score). */
int simseg_attention_fwd_qscaled(const void* qkv, void* out, float* lse, int64_t B, int64_t T, int64_t H, void* stream);
/* Backward: dqkv[B,T,3,H,64] from qkv, ctx, dctx and lse, all in `dtype` (0 = fp32: the exact mode of a non-AMP run,
 * simseg/core/hooks/optimizer.py:76-77; 1 = bf16).  workspace: caller-provided fp32 scratch of simseg_attention_bwd_workspace_bytes(B, T, H).
 * dqkv_colsum (optional, [3*H*64] fp32) += column sums of dqkv over all B*T rows: the bias gradient of the q/k/v projection (timm
 * Attention.qkv.bias, HF BertSelfAttention.{query,key,value}.bias), formed inside the one-kernel bf16 backward (T <= 256) and by a
 * column-sum pass over dqkv otherwise. */
int64_t simseg_attention_bwd_workspace_bytes(int64_t B, int64_t T, int64_t H);
int simseg_attention_bwd(const void* qkv, const int64_t* key_mask, const void* out, const void* dout, const float* lse,
                         float* workspace, void* dqkv, float* dqkv_colsum, int dtype, int64_t B, int64_t T, int64_t H, float scale,
                         uint64_t drop_seed, float drop_p, int skip_padded_rows, void* stream);

/* The same attention for a ragged batch stored WITHOUT its padding (bf16, at most T <= 256 tokens per sequence): sequence b is rows
 * [row_start[b], row_start[b+1]) of qkv [rows,3,H,64] / out, dout [rows,H*64] / dqkv (row_start: B+1 ascending int32 on the device),
 * every stored token is real.  Attention has no notion of position, so this IS HF BertSelfAttention with its key-padding mask
 * (huggingface_builder.py:16-17) evaluated on the unmasked tokens; the rows a padded layout would carry are never formed.  lse
 * [B,H,T] (log2 domain).  Rows outside every sequence (a caller's tile padding) are not touched.  workspace / dqkv_colsum as in
 * simseg_attention_bwd. */
int simseg_attention_fwd_rows(const void* qkv, const int32_t* row_start, void* out, float* lse, int64_t B, int64_t T, int64_t H, float scale,
                              uint64_t drop_seed, float drop_p, void* stream);
int simseg_attention_bwd_rows(const void* qkv, const int32_t* row_start, const void* out, const void* dout, const float* lse,
                              float* workspace, void* dqkv, float* dqkv_colsum, int64_t B, int64_t T, int64_t H, float scale, uint64_t drop_seed,
                              float drop_p, void* stream);

/* The same attention on PLANE-MAJOR projection operands (round 4; 16-bit, T <= 256): qkv / dqkv are [3*H][plane_rows][64] - plane
 * which*H + h holds the 64 channels of head h of q (which = 0), k (1) or v (2) of every token row, i.e. the packed projection
 * [rows,3,H,64] with the (3,H) axes moved in front of the rows.  It is what simseg_gemm writes with c_planes = plane_rows and reads with
 * a_planes (the A operand's K-tiles are the planes), so the layout never leaves the three kernels; a head's operand rows are one
 * contiguous run instead of T cache lines 6*H*64 bytes apart.  Sequence b = rows [b*T, (b+1)*T) when row_start is NULL (timm
 * Attention.forward, vit_builder.py:18), else [row_start[b], row_start[b+1]) as in simseg_attention_fwd_rows (HF BertSelfAttention,
 * huggingface_builder.py:16-17).  out / dout [rows,H*64], lse [B,H,T], workspace / dqkv_colsum as in simseg_attention_bwd. */
int simseg_attention_fwd_planes(const void* qkv, int64_t plane_rows, const int32_t* row_start, void* out, float* lse, int64_t B, int64_t T,
                                int64_t H, float scale, uint64_t drop_seed, float drop_p, void* stream);
int simseg_attention_bwd_planes(const void* qkv, int64_t plane_rows, const int32_t* row_start, const void* out, const void* dout,
                                const float* lse, float* workspace, void* dqkv, float* dqkv_colsum, int64_t B, int64_t T, int64_t H,
                                float scale, uint64_t drop_seed, float drop_p, void* stream);

/* Prompt ensemble of the zero-shot classifier: out[s,:] = normalize(mean over the P prompt embeddings x[s,:,:]).
 * tools/seg_evaluation.py:71-73 (class_embeddings.mean(dim=0); /= norm()). */
int simseg_segment_mean_l2norm(const float* x, float* out, int64_t S, int64_t P, int64_t D, void* stream);

/* rnorm[r] = 1 / max(||x_r||_2, eps): the F.normalize of tools/seg_evaluation.py:112 as a GEMM row scale. */
int simseg_row_rnorm(const void* x, int dtype, float* rnorm, int64_t rows, int64_t D, float eps, void* stream);

/* InfoNCE rows over sims[N1,N2] = feat1 . feat2_global^T (fp32, from simseg_gemm): z = s / clamp(T,1e-3,0.5),
 * per-row cross-entropy against target column target0 + i with optional label smoothing and ignore weights, top-1
 * hit, and (write_grad) dLoss/ds written IN PLACE over sims.  out3 = {loss, top-1 acc, dLoss/dT}.
 * simseg/models/criteria/losses/mml_loss.py:56,73-77,89-95 (+ :350-376 LabelSmoothingCrossEntropy,
 * simseg/utils/misc.py:462-478 calc_topk_accuracy). */
int simseg_nce_rows(float* sims, const float* temperature, const float* ignore_mask, float* row_loss, float* row_correct,
                    float* row_tdot, float* out3, int64_t N1, int64_t N2, int64_t target0, float smoothing, int write_grad,
                    void* stream);
/* y[r,:] = alpha * x[r,:] * (one_minus ? 1 - s[r] : s[r])  -- feat2_global * (1 - ignore_mask), mml_loss.py:70-71. */
int simseg_scale_rows(const float* x, const float* s, float* y, int64_t rows, int64_t D, int one_minus, float alpha, void* stream);

/* y[i] = alpha * scalar[0] * x[i] with the scalar read on the device (upstream loss gradient, no host sync). */
int simseg_scale_by_scalar(const float* x, const float* scalar, float* y, int64_t n, float alpha, void* stream);

/* Both directions of the CLIP loss (simseg/models/pipelines/clip.py:129-140: 0.5 * (NCE(image, text) + NCE(text, image)), each as
 * simseg_nce_rows without an ignore mask) in two launches: sims2 = the two [N1, N2] similarity blocks stacked (image -> text first), overwritten
 * by their gradients w.r.t. the similarities when write_grad; row_scratch: 6 * N1 floats; out4 = {loss, i2t top-1 acc, t2i top-1 acc,
 * dLoss/dTemperature}. */
int simseg_nce_pair(float* sims2, const float* temperature, float* row_scratch, float* out4, int64_t N1, int64_t N2, int64_t target0,
                    float smoothing, int write_grad, void* stream);
/* n <= 6 fp32 transposes in ONE launch (out[k][c, r] = in[k][r, c], in[k] [rows[k], cols[k]] contiguous); jobs with scale[k] != 0 are
 * multiplied by alpha * scalar[0] - in the transposed copy and IN PLACE; y0[0] = scalar[0] * x0[0] when y0 is given.  The loss head's
 * backward (mml_loss.py:73 differentiated) needs dS, dS^T and the transposed embeddings as row . row GEMM operands, all times the upstream
 * gradient: one launch instead of the round-3 head's four transposes and three scale passes per direction. */
int simseg_transpose_multi(const void* const* in, void* const* out, const int64_t* rows, const int64_t* cols, const int32_t* scale, int64_t n,
                           const float* scalar, float alpha, const float* x0, float* y0, void* stream);
/* Retrieval: rank[i] = #{j : sim_ij > max_{j': gid match} sim_ij'}, has_match[i] = any gid match.  Equals the
 * argsort/gather/first-match rank of simseg/tasks/clip/hooks/utils.py:36-42,64-66 on tie-free scores. */
int simseg_retrieval_rank(const float* sim, const int64_t* left_gid, const int64_t* right_gid, int32_t* has_match, int32_t* rank,
                          int64_t M, int64_t N, int64_t ld, void* stream);
/* The reverse direction from the SAME matrix: column j retrieves rows (best = max over rows with row_gid == col_gid[j], rank = rows
 * scoring strictly higher) - two row-major passes instead of a second GEMM on swapped operands.  scratch: N int32. */
int simseg_retrieval_rank_cols(const float* sim, const int64_t* row_gid, const int64_t* col_gid, int32_t* has_match, int32_t* rank,
                               int32_t* scratch, int64_t M, int64_t N, int64_t ld, void* stream);
/* counts4 = {#has_match, #rank<b0, #rank<b1, #rank<b2}  (hooks/utils.py:69-71). */
int simseg_recall_counts(const int32_t* has_match, const int32_t* rank, int64_t M, int b0, int b1, int b2, int32_t* counts4,
                         void* stream);

/* torch.optim.AdamW (configs/clip/simseg.vit-b.yaml:31-36) over a flat fp32 segment; refreshes the bf16 compute copy. */
int simseg_adamw_step(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr, float beta1, float beta2,
                      float eps, float weight_decay, int64_t step, float grad_scale, void* stream);

/* Same update for every parameter tensor in ONE launch.  table[t] = six 8-byte words {p, g, m, v, p16, (float lr, float wd)}: five
 * device pointers (p16 = bf16 compute copy to refresh, or 0) and the tensor's own learning rate / weight decay packed as two floats
 * (the reference's ClipOptimizerHook makes one param group per parameter, tasks/clip/hooks/optimizer.py:18-36); sizes[t] elements;
 * chunk c covers [chunk_off[c], chunk_off[c]+chunk) of tensor chunk_tid[c]. */
int simseg_adamw_multi_step(const void* table, const int64_t* sizes, const int32_t* chunk_tid, const int64_t* chunk_off,
                            int64_t n_chunks, int chunk, float beta1, float beta2, float eps, int64_t step, float grad_scale,
                            void* stream);

/* The same launch inside the reference's fp16 AMP iteration (clip_runner.py:226-230; core/hooks/optimizer.py:73-82: scaler.scale(loss)
 * .backward(), scaler.step(optimizer), scaler.update()) WITHOUT the host read torch's scaler.step makes for an optimizer that cannot skip
 * a step by itself: loss_scale / found_inf are torch.amp.GradScaler's device tensors (its `optimizer.grad_scale` / `optimizer.found_inf`
 * contract; either may be null).  Gradients are used as g * grad_scale / loss_scale[0]; when found_inf[0] != 0 nothing is updated.  The
 * count of steps actually taken lives on the device: step_in[0] is read, step_out[0] = step_in[0] + (skipped ? 0 : 1) is written (two
 * distinct slots) and supplies the bias corrections. */
int simseg_adamw_multi_step_amp(const void* table, const int64_t* sizes, const int32_t* chunk_tid, const int64_t* chunk_off,
                                int64_t n_chunks, int chunk, float beta1, float beta2, float eps, float grad_scale,
                                const float* loss_scale, const float* found_inf, const float* step_in, float* step_out, void* stream);
/* found_inf[0] = 1 if any gradient element addressed by the table (same layout as above; only the g pointers are read) is inf / nan,
 * else unchanged: GradScaler's overflow check (torch._amp_foreach_non_finite_check_and_unscale_ with inverse scale 1) as one read-only
 * launch; the unscaling rides on simseg_adamw_multi_step_amp. */
int simseg_grads_nonfinite(const void* table, const int64_t* sizes, const int32_t* chunk_tid, const int64_t* chunk_off, int64_t n_chunks,
                           int chunk, float* found_inf, void* stream);

int simseg_cast(const void* in, void* out, int64_t n, int to_bf16, void* stream);
int simseg_transpose_f32(const float* in, float* out, int64_t R, int64_t C, void* stream);
/* fp32 [rows, K] (row stride ld_in) -> bf16 [rows, 6 K]: the three round-to-nearest bf16 pieces hi / mid / lo of every element
 * (hi + mid + lo == x), laid out along K as [hi|hi|hi|mid|mid|lo] (b_pattern = 0, the A operand) or [hi|mid|lo|hi|mid|hi] (b_pattern = 1,
 * the B operand; b_pattern = 2: three planes bf16 [3][rows, K] instead, the operand form of simseg_attention_fwd_x3), so that ONE simseg_gemm call on the two bf16 images forms the six leading piece products of the fp32 product in its
 * fp32 accumulators - the exact-mode nn.Linear of the evaluation tools (torch fp32 matmul, vit_builder.py:18 / huggingface_builder.py:16-17
 * without autocast) at the bf16 MFMA rate; the three dropped products are <= 2^-24 |a||b|, the size of an fp32 FMA's own rounding. */
int simseg_split_bf16x3(const float* in, void* out, int64_t rows, int64_t K, int64_t ld_in, int b_pattern, void* stream);
/* Exact-mode attention forward (timm Attention.forward in fp32, vit_builder.py:18, as the evaluation tools run it) on the bf16 matrix pipe:
 * qkv3 = the three bf16 pieces of the fp32 packed projection as planes (simseg_split_bf16x3 with b_pattern = 2: bf16 [3][B, T, 3, H, 64],
 * piece p at qkv3 + p * plane_elems); scores and outputs accumulate the six leading piece products in fp32 (fp32-grade accuracy, like
 * the split GEMM).  out fp32 [B, T, H*64].  No mask / dropout / log-sum-exp: evaluation only. */
int simseg_attention_fwd_x3(const void* qkv3, int64_t plane_elems, float* out, int64_t B, int64_t T, int64_t H, float scale, void* stream);
/* dst[i,:] = src[idx[i],:] (idx[i] < 0: a zero row); rows of row_bytes (a multiple of 16) bytes, any dtype.  Drops / restores the padded
 * token rows of ragged caption batches around the text tower's GEMMs (HF BertModel computes them: huggingface_builder.py:16-17). */
int simseg_gather_rows(const void* src, const int32_t* idx, void* dst, int64_t n, int64_t row_bytes, void* stream);
/* Row maps of a ragged caption batch from its 0/1 attention mask [B, L] (int64, as the reference's tokenizer delivers it:
 * simseg/datasets/clip/clip_dataset.py:127-133), in one launch and without a host read: idx int32 [cap] = flat positions b*L+l of the real
 * tokens in raster order, then -1 up to the next multiple of `multiple` (or cap); inv int32 [B*L] = packed row of every position (-1 at
 * padded positions); row_start int32 [B+1]; info int32 [4] = {real tokens, 1 if a sequence has a real token behind a padded one,
 * 1 if they did not fit cap rows or differ from `expect` (>= 0: the count the caller sized its buffers for, e.g. from the loader's
 * caption lengths; -1: unknown), rows of idx written}.  What the text tower needs to skip the padded token rows HF's BertModel computes
 * (huggingface_builder.py:16-17) - replaces ~15 torch index kernels and two host reads per step.  B <= 8192. */
int simseg_ragged_maps(const int64_t* mask, int64_t B, int64_t L, int64_t multiple, int64_t cap, int64_t expect, int32_t* idx, int32_t* inv,
                       int32_t* row_start, int32_t* info, void* stream);
/* g[i] = keep(seed, i) ? g[i] / (1-p) : 0 -- regenerates the forward dropout mask of simseg_gemm for the backward. */
int simseg_dropout_apply(void* g, int dtype, int64_t n, uint64_t seed, float p, void* stream);

/* ---- zero-shot segmentation post-processing (tools/seg_evaluation.py:112-170) ---------------------------------
 * scores [B,C] = pooled . text^T.  Top `top_cls_num` classes (:121), threshold = mean + unbiased std of those (:122-123); the
 * first `ncand` (reference: 5, :128-130) become candidates.  cand_idx[b,i] = class index, or -1 where the reference skips the
 * slot (class 0 / 255 :132-133, or score < threshold :145-146); cand_score[b,i] = its score; threshold[b] optional. */
int simseg_seg_select(const float* scores, int* cand_idx, float* cand_score, float* threshold, int64_t B, int64_t C,
                      int64_t top_cls_num, int64_t ncand, void* stream);
/* sim [B, n*n, C] fp32 (simseg_patch_text_sim).  Per valid candidate: its column, min-max normalised (:148-149) -> prob
 * [B,ncand,n*n] (optional, the DenseCRF unary input) and the binary map prob > 0.5 scaled to 255 (what dense_crf :30-54 returns
 * with its pairwise terms off), nearest-upsampled x16 (:137) -> mask [B,ncand,16n,16n] bytes.  Invalid slots are not written. */
int simseg_seg_masks(const float* sim, const int* cand_idx, float* prob, void* mask, int64_t B, int64_t n, int64_t C, int64_t ncand,
                     void* stream);
/* The same on a rectangular patch grid: sim [B, nh*nw, C] (row-major patch rows), mask [B,ncand,16nh,16nw] - the stitched map of a
 * sliding-window evaluation (BASELINE configs[3]); simseg_seg_masks is the nh = nw case. */
int simseg_seg_masks_rect(const float* sim, const int* cand_idx, float* prob, void* mask, int64_t B, int64_t nh, int64_t nw, int64_t C,
                          int64_t ncand, void* stream);
/* Sliding-window stitch (BASELINE configs[3] / SURVEY.md 8d cfg 4: 512x512 windows at stride 256 over a larger image, "overlap-average
 * sim maps"; the reference tool itself resizes to one window, tools/seg_evaluation.py:84-85,109).  win [B, wy, wx, n*n, C] fp32 = the
 * per-window similarity maps (simseg_patch_text_sim), window (i, j) at patch offset (i*step, j*step); out [B, nh*nw, C] with
 * nh = n + (wy-1)*step, nw = n + (wx-1)*step: per cell the mean over the covering windows, summed in (i, j) order.  step = 0: the plain
 * mean over all windows (with n = 1: image-level class scores from per-window scores). */
int simseg_stitch_windows(const float* win, float* out, int64_t B, int64_t wy, int64_t wx, int64_t n, int64_t step, int64_t C, void* stream);
/* cv2.dilate / cv2.erode with a 7x7 ones kernel, ONE iteration (the third positional argument in :156-157 is `dst`, not
 * `iterations`), default border (never wins) on byte images [M,H,W]; erode = 0 dilate, 1 erode.  out must not alias in. */
int simseg_morph7(const void* in, void* out, int64_t M, int64_t H, int64_t W, int erode, void* stream);
/* cv2.dilate followed by cv2.erode (:156-157) in one pass over byte images [M,H,W] (the dilated image never leaves LDS); same
 * border rules as simseg_morph7 applied twice.  valid (optional, [M] ints): images with valid[m] < 0 are skipped and their
 * output left untouched (candidate slots the reference never visits). */
int simseg_close7(const void* in, void* out, const int* valid, int64_t M, int64_t H, int64_t W, void* stream);
/* masks [B,ncand,Hm,Wm] bytes -> nearest resize to (H,W) (cv2.INTER_NEAREST :159) -> temp_pred[class] = mask * score (:160) ->
 * argmax over classes (:163; first maximum, class 0 when nothing is positive) -> pred [B,H,W] int32 (optional) and
 * hist [3,C] uint64 += {intersect, pred area, label area} over pixels whose label != ignore_index (utils/metrics.py:60-74).
 * labels [B,H,W] bytes. */
int simseg_seg_predict(const void* masks, const int* cand_idx, const float* cand_score, const void* labels, int* pred, void* hist,
                       int64_t B, int64_t ncand, int64_t Hm, int64_t Wm, int64_t H, int64_t W, int64_t C, int64_t ignore_index,
                       void* stream);

/* The fully connected CRF the reference applies to every visited candidate map (tools/seg_evaluation.py:31-54, called at :153:
 * pydensecrf DenseCRF2D with 2 labels, U = -log([1-p, p] + 1e-8), addPairwiseGaussian(sxy=3, compat=3), addPairwiseBilateral(sxy=40,
 * srgb=13, rgbim, compat=10), inference(3), argmax), as mean-field inference on two permutohedral lattices built on the device.
 * rgb [B,H,W,3] bytes (the de-normalised network inputs, RGB order), prob [B,C,H,W] fp32 in [0,1] (C <= 8 candidate maps per image:
 * an image's maps share its lattices; the images of a batch are independent problems solved side by side in the same launches) ->
 * mask [B,C,H,W] bytes (255 where the pixel is labelled as the class); q_out (optional) [B,C,H,W] fp32 = Q(label 1) after the last
 * iteration.  workspace: caller-allocated, 256-byte aligned, simseg_dense_crf_workspace_bytes(B, H, W, C).  reuse_spatial != 0: the caller
 * vouches that the previous call on this workspace had the same H, W and sxy_g and that the workspace is untouched since - the spatial
 * (Gaussian) lattice, which depends on nothing else, is then not rebuilt. */
int64_t simseg_dense_crf_workspace_bytes(int64_t B, int64_t H, int64_t W, int64_t C);
int simseg_dense_crf(const uint8_t* rgb, const float* prob, uint8_t* mask, float* q_out, int64_t B, int64_t C, int64_t H, int64_t W, float sxy_g,
                     float compat_g, float sxy_b, float srgb, float compat_b, int iters, void* workspace, int64_t workspace_bytes,
                     int reuse_spatial, void* stream);

/* debug: bf16 attention forward that also writes a 5-entry cycle-counter timeline of block (0,0) to dbg (tools/dbg_attn_timeline.py). */
int simseg_debug_attention_timeline(const void* qkv, void* out, float* lse, void* dbg, int64_t B, int64_t T, int64_t H, void* stream);

/* hardware probe used by tests: lane/element map of ds_read_b64_tr_b16. */
int simseg_debug_tr16_probe(int* out_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif
