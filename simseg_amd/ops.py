"""Tensor-level wrappers over the C ABI (include/simseg_hip.h).  PyTorch supplies device memory and the stream;
all arithmetic runs in libsimseg_hip.so.  No autograd here (see autograd.py) and no CPU fallback."""
import os
import threading

import torch

from .lib import call, ptr, require_gpu, stream, raw

F32, BF16 = 0, 1
_DT = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: BF16}       # code 1 = "the 16-bit type": bf16, or fp16 while the call's tensors are fp16 (lib.call)
HALF_TYPES = (torch.bfloat16, torch.float16)


def dt(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"simseg_amd supports fp32, bf16 and fp16 tensors, got {t.dtype}")


def _c(t):
    if t is not None and not t.is_contiguous():
        raise ValueError("simseg_amd ops need contiguous tensors")
    return t


# The library's kernel selectors are thread-local (include/simseg_hip.h: the compute entry points are re-entrant), and autograd runs
# backward on its own threads: the selection made here is kept in Python and pushed to whichever thread launches next.
_VARIANT = {"gemm": int(os.environ.get("SIMSEG_GEMM_VARIANT", "0")), "attention": int(os.environ.get("SIMSEG_ATTN_VARIANT", "0"))}     # (env: A/B runs)
_PUSHED = threading.local()


def _push_variant(kind):
    want = _VARIANT[kind]
    if getattr(_PUSHED, kind, 0) != want:
        call(f"simseg_set_{kind}_variant", want)
        setattr(_PUSHED, kind, want)


def set_gemm_variant(v):
    """0 auto, 1 = 128x128 register-staged kernel, 2 = 256x256 direct-to-LDS ring, 3 = 256x256 ping-pong, 4 = small-problem kernel wherever it applies (tests / benchmarks only)."""
    _VARIANT["gemm"] = int(v)
    _push_variant("gemm")


def set_attention_variant(v):
    """0 auto (bf16 sequences of <= 256 tokens on the resident kernels, their backward as one kernel; unmasked 16-bit sequences of >= 512
    tokens on the 64-queries-per-wave forward, small launches on its one-query-block form), 1 = always the streaming ring kernels, 3 = backward
    as the two resident passes, 6 / 7 = the long-sequence forward with one / two query blocks per wave whatever the launch size, 8 = auto
    with the long-sequence forward held to ONE block per CU (the occupancy probe tools/scratch/attn_w64_occ.py)  (tests / benchmarks only)."""
    _VARIANT["attention"] = int(v)
    _push_variant("attention")


PROFILE = None   # bench.py sets this to a list to time every GEMM launch with events on the launch stream


def gemm(a, b, *, trans_a=False, trans_b=False, out=None, out_dtype=None, alpha=1.0, bias=None, rowscale=None,
         residual=None, act=0, aux=None, aux_out=None, row_group=0, res_mod=False, accumulate=False, splitk=1,
         drop_seed=0, drop_p=0.0, out_rows=None, colsum=None):
    """C = epilogue(alpha * op(a) @ op(b)); a: [M,K] ([K,M] if trans_a); b: [N,K] ([K,N] if trans_b)."""
    require_gpu(a, b)
    _c(a); _c(b)
    M, K = (a.shape[1], a.shape[0]) if trans_a else (a.shape[0], a.shape[1])
    N = b.shape[1] if trans_b else b.shape[0]
    kb = b.shape[0] if trans_b else b.shape[1]
    if kb != K:
        raise ValueError(f"gemm: inner dims differ ({K} vs {kb})")
    if act in (7, 8):       # the 8-bit tile-blocked derivative image (include/simseg_hip.h): an opaque [M, N] byte tensor
        t8 = aux_out if act == 7 else aux
        if t8 is None or t8.dtype != torch.uint8 or t8.numel() < a.shape[0] * (b.shape[1] if trans_b else b.shape[0]):
            raise TypeError("gemm: act 7 / 8 take the derivative image as a uint8 tensor of M x N bytes")
    if a.dtype != b.dtype or (aux is not None and act != 8 and aux.dtype not in (a.dtype, torch.float32)):
        raise TypeError(f"gemm: operands of different types ({a.dtype}, {b.dtype}{'' if aux is None else ', aux ' + str(aux.dtype)}): bf16 and fp16 do not mix in one call")
    if out is None:
        rows = out_rows if out_rows is not None else M
        out = torch.empty(rows, N, device=a.device, dtype=out_dtype or a.dtype)
    _c(out)
    ldr = residual.shape[-1] if residual is not None else 0
    _push_variant("gemm")
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    call("simseg_gemm", ptr(a), ptr(b), ptr(out), M, N, K, a.shape[1], b.shape[1], out.shape[-1], dt(a), dt(out),
         int(trans_a), int(trans_b), float(alpha), ptr(_c(bias)), ptr(_c(rowscale)), ptr(_c(residual)), ldr, int(act),
         ptr(_c(aux)), ptr(_c(aux_out)), int(row_group), int(res_mod), int(accumulate), int(splitk), int(drop_seed),
         float(drop_p), ptr(colsum), stream())
    if PROFILE is not None:
        e1.record()
        kind = ("f32" if a.dtype == torch.float32 else "bf16") + "_" + ("t" if trans_a else "n") + ("n" if trans_b else "t")
        kind += "_o32" if out.dtype == torch.float32 else "_o16"
        kind += {1: "", 2: "_L", 3: "_P", 4: "_S", 5: "_R", 8: "_X", 9: "_Y", 10: "_Q"}[raw("simseg_gemm_last_variant")]      # the kernel the library actually launched
        PROFILE.append((kind, 2.0 * M * N * K, e0, e1))
    return out


_BLK_OK = {}


def gemm_aux_blocked_ok(M, N, K):
    """May the fc1 forward save GELU' as the tile-blocked accumulator image (act 5) for the dgrad through fc2 (act 6)?  (include/simseg_hip.h:
    full 256x256 tiles on the ping-pong kernels.)  SIMSEG_AMD_BLOCKED_AUX=0 keeps the row-major tensor (A/B runs)."""
    key = (int(M), int(N), int(K))
    r = _BLK_OK.get(key)
    if r is None:
        r = _BLK_OK[key] = os.environ.get("SIMSEG_AMD_BLOCKED_AUX", "1") != "0" and bool(raw("simseg_gemm_aux_blocked_ok", *key))
    return r


def layernorm_fwd(x, gamma, beta, eps, out_dtype=torch.float32, want_bf16_copy=False, save_stats=False):
    require_gpu(x)
    D = x.shape[-1]
    rows = x.numel() // D
    y = torch.empty(x.shape, device=x.device, dtype=out_dtype)
    # (want_bf16_copy: False, True = a bf16 copy, or the 16-bit dtype the copy should have)
    y16 = torch.empty(x.shape, device=x.device, dtype=want_bf16_copy if isinstance(want_bf16_copy, torch.dtype) else torch.bfloat16) if want_bf16_copy else None
    mean = torch.empty(rows, device=x.device, dtype=torch.float32) if save_stats else None
    rstd = torch.empty(rows, device=x.device, dtype=torch.float32) if save_stats else None
    call("simseg_layernorm_fwd", ptr(_c(x)), ptr(gamma), ptr(beta), ptr(y), dt(y), ptr(y16), ptr(mean), ptr(rstd), rows, D,
         float(eps), stream())
    return y, y16, mean, rstd


def layernorm_bwd(x, mean, rstd, gamma, dgamma, dbeta, dy16=None, dy32=None, dres=None, want_f32=True, want_bf16=True,
                  dxsum=None, drop_seed=0, drop_p=0.0, dres16=None, y16=None, beta=None):
    require_gpu(x)
    D = x.shape[-1]
    rows = x.numel() // D
    dx32 = torch.empty(x.shape, device=x.device, dtype=torch.float32) if want_f32 else None
    h16 = dy16.dtype if dy16 is not None else (want_bf16 if isinstance(want_bf16, torch.dtype) else torch.bfloat16)
    dx16 = torch.empty(x.shape, device=x.device, dtype=h16) if want_bf16 else None
    partials = torch.empty(raw("simseg_layernorm_bwd_workspace_bytes", rows, D) // 4, device=x.device, dtype=torch.float32)
    call("simseg_layernorm_bwd", ptr(_c(dy16)), ptr(_c(dy32)), ptr(_c(dres)), ptr(_c(dres16)), ptr(_c(x)), ptr(_c(y16)), ptr(_c(beta)) if y16 is not None else None,
         ptr(mean), ptr(rstd), ptr(gamma),
         ptr(dx32), ptr(dx16), ptr(dgamma), ptr(dbeta), ptr(dxsum), ptr(partials), rows, D, int(drop_seed), float(drop_p), stream())
    return dx32, dx16


def colsum_accum(x2d, out):
    require_gpu(x2d)
    call("simseg_colsum_accum", ptr(_c(x2d)), dt(x2d), ptr(out), x2d.shape[0], x2d.shape[1], x2d.shape[1], stream())
    return out


def vit_im2col(image, out_dtype):
    require_gpu(image)
    B, C, H, W = image.shape
    if C != 3 or image.dtype != torch.float32:
        raise ValueError("vit_im2col expects fp32 [B,3,H,W]")
    cols = torch.empty(B * (H // 16) * (W // 16), 768, device=image.device, dtype=out_dtype)
    call("simseg_vit_im2col", ptr(_c(image)), ptr(cols), dt(cols), B, H, W, stream())
    return cols


def vit_cls_rows(cls, pos, x):
    B, T, D = x.shape
    call("simseg_vit_cls_rows", ptr(cls), ptr(pos), ptr(x), B, T, D, stream())


def vit_cls_grad(dx, dcls):
    B, T, D = dx.shape
    call("simseg_vit_cls_grad", ptr(_c(dx)), ptr(dcls), B, T, D, stream())


def bert_embed_fwd(ids, word, pos, type_emb):
    require_gpu(ids, word)
    B, L = ids.shape
    D = word.shape[1]
    out = torch.empty(B, L, D, device=word.device, dtype=torch.float32)
    call("simseg_bert_embed_fwd", ptr(_c(ids)), ptr(word), ptr(pos), ptr(type_emb), ptr(out), B, L, D, word.shape[0], stream())
    return out


def bert_embed_bwd(ids, mask, dsum, dword):
    B, L = ids.shape
    call("simseg_bert_embed_bwd", ptr(_c(ids)), ptr(_c(mask)), ptr(_c(dsum)), ptr(dword), B, L, dsum.shape[-1], dword.shape[0], stream())


def topk_pool_l2norm_fwd(tok, k, mask=None, eps=1e-8, normalize=True):
    require_gpu(tok)
    B, N, P = tok.shape
    emb = torch.empty(B, P, device=tok.device, dtype=torch.float32)
    idx = torch.empty(B, k, P, device=tok.device, dtype=torch.int32)
    norm = torch.empty(B, device=tok.device, dtype=torch.float32)
    scratch = None
    if B <= 256 and N >= 256:        # long token axes: scan 32 token slices per image in parallel, then merge (tools/pool_bench.py, N = 1024: 141 vs
                                     # 522 us at B = 63, 244 vs 600 us at B = 128; one block per image walks its tokens serially)
        scratch = torch.empty(raw("simseg_topk_pool_workspace_bytes", B, P, int(k)) // 4, device=tok.device, dtype=torch.float32)
    call("simseg_topk_pool_l2norm_fwd", ptr(_c(tok)), dt(tok), ptr(_c(mask)), ptr(emb), ptr(idx), ptr(norm), ptr(scratch), B, N, P, int(k),
         float(eps), int(normalize), stream())
    return emb, idx, norm


def topk_pool_l2norm_bwd(demb, emb, norm, idx, N, dtype, eps=1e-8, normalize=True):
    B, k, P = idx.shape
    dtok = torch.empty(B, N, P, device=emb.device, dtype=dtype)
    call("simseg_topk_pool_l2norm_bwd", ptr(_c(demb)), ptr(emb), ptr(norm), ptr(idx), ptr(dtok), dt(dtok), B, N, P, int(k), float(eps), int(normalize), stream())
    return dtok


def attention_fwd(qkv, heads, mask=None, scale=0.125, save_lse=False, drop_seed=0, drop_p=0.0, skip_padded_rows=False, zero_skipped=True):
    """qkv [B,T,3*H*64] packed (3,H,64) -> ctx [B,T,H*64].
    skip_padded_rows: the kernels do not write the rows past a sequence's last unmasked key (include/simseg_hip.h).  Those rows of the
    returned tensors are zeros unless the caller passes zero_skipped=False because it provably never reads them (the packed text
    tower gathers the real rows only and saves a fill pass per layer)."""
    require_gpu(qkv)
    B, T, W = qkv.shape
    if W != 3 * heads * 64:
        raise ValueError("attention: qkv last dim must be 3*heads*64")
    alloc = torch.zeros if (skip_padded_rows and zero_skipped) else torch.empty
    out = alloc(B, T, heads * 64, device=qkv.device, dtype=qkv.dtype)
    lse = alloc(B, heads, T, device=qkv.device, dtype=torch.float32) if save_lse else None
    _push_variant("attention")
    call("simseg_attention_fwd", ptr(_c(qkv)), ptr(_c(mask)), ptr(out), ptr(lse), dt(qkv), B, T, heads, float(scale),
         int(drop_seed), float(drop_p), int(skip_padded_rows), stream())
    return out, lse


def attention_bwd(qkv, out, dout, lse, heads, mask=None, scale=0.125, drop_seed=0, drop_p=0.0, skip_padded_rows=False, colsum=None,
                  zero_skipped=True):
    """dqkv; colsum (optional fp32 [3*H*64]) += column sums of dqkv, the q/k/v bias gradient (formed inside the kernel for T <= 256).
    skip_padded_rows / zero_skipped: as in attention_fwd - the skipped rows of dqkv are zeros (their true gradient) unless the caller
    never reads them."""
    B, T, W = qkv.shape
    dqkv = torch.zeros_like(qkv) if (skip_padded_rows and zero_skipped) else torch.empty_like(qkv)
    ws = torch.empty(raw("simseg_attention_bwd_workspace_bytes", B, T, heads) // 4, device=qkv.device, dtype=torch.float32)
    _push_variant("attention")
    call("simseg_attention_bwd", ptr(_c(qkv)), ptr(_c(mask)), ptr(_c(out)), ptr(_c(dout)), ptr(lse), ptr(ws), ptr(dqkv), ptr(colsum),
         dt(qkv), B, T, heads, float(scale), int(drop_seed), float(drop_p), int(skip_padded_rows), stream())
    return dqkv


def attention_fwd_rows(qkv, heads, row_start, max_len, scale=0.125, save_lse=False, drop_seed=0, drop_p=0.0, n_real=None):
    """Ragged batch without padding: qkv [rows, 3*H*64] bf16, sequence b = rows [row_start[b], row_start[b+1]) (int32 [B+1]), at most
    max_len tokens each -> ctx [rows, H*64] (rows outside every sequence - a tile padding behind row n_real - are zeros), lse [B,H,max_len]."""
    require_gpu(qkv, row_start)
    if qkv.dtype not in HALF_TYPES or row_start.dtype != torch.int32:
        raise TypeError("attention_fwd_rows: bf16 qkv, int32 row_start")
    rows, W = qkv.shape
    B = row_start.numel() - 1
    out = torch.empty(rows, heads * 64, device=qkv.device, dtype=qkv.dtype)
    if n_real is not None and n_real < rows:
        out[n_real:].zero_()
    lse = torch.empty(B, heads, max_len, device=qkv.device, dtype=torch.float32) if save_lse else None
    _push_variant("attention")
    call("simseg_attention_fwd_rows", ptr(_c(qkv)), ptr(_c(row_start)), ptr(out), ptr(lse), B, int(max_len), heads, float(scale), int(drop_seed),
         float(drop_p), stream())
    return out, lse


def attention_bwd_rows(qkv, out, dout, lse, heads, row_start, max_len, scale=0.125, drop_seed=0, drop_p=0.0, colsum=None, n_real=None):
    """dqkv [rows, 3*H*64] for attention_fwd_rows (rows behind n_real zeroed); colsum (optional fp32 [3*H*64]) += its column sums."""
    rows, W = qkv.shape
    B = row_start.numel() - 1
    dqkv = torch.empty_like(qkv)
    if n_real is not None and n_real < rows:
        dqkv[n_real:].zero_()
    ws = torch.empty(raw("simseg_attention_bwd_workspace_bytes", B, int(max_len), heads) // 4, device=qkv.device, dtype=torch.float32)
    _push_variant("attention")
    call("simseg_attention_bwd_rows", ptr(_c(qkv)), ptr(_c(row_start)), ptr(_c(out)), ptr(_c(dout)), ptr(lse), ptr(ws), ptr(dqkv), ptr(colsum),
         B, int(max_len), heads, float(scale), int(drop_seed), float(drop_p), stream())
    return dqkv


def attention_fwd_planes(qkvp, heads, B, T, row_start=None, scale=0.125, save_lse=False, drop_seed=0, drop_p=0.0, n_real=None):
    """Plane-major operands: qkvp [3*H, rows, 64] (16-bit) -> ctx [rows, H*64], lse [B,H,T].  Sequence b = rows [b*T, (b+1)*T) (row_start None)
    or [row_start[b], row_start[b+1]) (int32 [B+1]); rows outside every sequence (behind n_real) are zeros."""
    require_gpu(qkvp)
    P3, rows, d = qkvp.shape
    if qkvp.dtype not in HALF_TYPES or P3 != 3 * heads or d != 64 or not qkvp.is_contiguous():
        raise ValueError("attention_fwd_planes: contiguous 16-bit [3*H, rows, 64]")
    out = torch.empty(rows, heads * 64, device=qkvp.device, dtype=qkvp.dtype)
    if n_real is not None and n_real < rows:
        out[n_real:].zero_()
    lse = torch.empty(B, heads, T, device=qkvp.device, dtype=torch.float32) if save_lse else None
    _push_variant("attention")
    call("simseg_attention_fwd_planes", ptr(qkvp), rows, ptr(_c(row_start)), ptr(out), ptr(lse), B, int(T), heads, float(scale), int(drop_seed),
         float(drop_p), stream())
    return out, lse


def attention_bwd_planes(qkvp, out, dout, lse, heads, B, T, row_start=None, scale=0.125, drop_seed=0, drop_p=0.0, colsum=None, n_real=None):
    """dqkv [3*H, rows, 64] for attention_fwd_planes (rows behind n_real zeroed); colsum (optional fp32 [3*H*64]) += its column sums."""
    P3, rows, d = qkvp.shape
    dqkv = torch.empty_like(qkvp)
    if n_real is not None and n_real < rows:
        dqkv[:, n_real:].zero_()
    ws = torch.empty(raw("simseg_attention_bwd_workspace_bytes", B, int(T), heads) // 4, device=qkvp.device, dtype=torch.float32)
    _push_variant("attention")
    call("simseg_attention_bwd_planes", ptr(qkvp), rows, ptr(_c(row_start)), ptr(_c(out)), ptr(_c(dout)), ptr(lse), ptr(ws), ptr(dqkv), ptr(colsum),
         B, int(T), heads, float(scale), int(drop_seed), float(drop_p), stream())
    return dqkv


def segment_mean_l2norm(x):
    """[S,P,D] fp32 -> [S,D]: unit-norm mean over P."""
    require_gpu(x)
    S, Pn, D = x.shape
    out = torch.empty(S, D, device=x.device, dtype=torch.float32)
    call("simseg_segment_mean_l2norm", ptr(_c(x)), ptr(out), S, Pn, D, stream())
    return out


def patch_text_sim(x2d, text, eps=1e-12, normalize=True):
    """Fused K14 kernel: x2d [M,K], text [C,K] (same dtype, C <= 256) -> fp32 [M,C] cosine map."""
    require_gpu(x2d, text)
    if x2d.dtype != text.dtype:
        raise TypeError("patch_text_sim: x and text must share a dtype")
    M, K = x2d.shape
    C = text.shape[0]
    out = torch.empty(M, C, device=x2d.device, dtype=torch.float32)
    call("simseg_patch_text_sim", ptr(_c(x2d)), ptr(_c(text)), ptr(out), M, C, K, dt(x2d), float(eps), int(normalize), stream())
    return out


def row_rnorm(x2d, eps=1e-12):
    require_gpu(x2d)
    rn = torch.empty(x2d.shape[0], device=x2d.device, dtype=torch.float32)
    call("simseg_row_rnorm", ptr(_c(x2d)), dt(x2d), ptr(rn), x2d.shape[0], x2d.shape[1], float(eps), stream())
    return rn


def cast(x, dtype, out=None):
    require_gpu(x)
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=dtype)
    call("simseg_cast", ptr(_c(x)), ptr(out), x.numel(), int(dtype in HALF_TYPES), stream())
    return out


ATTN_QSCALED_MIN_T = 512       # W64_MINT in csrc/attn.hip


def attention_qscale(scale=0.125):
    """The factor attention_fwd_qscaled expects on the q columns of the projection: softmax scale times log2(e)."""
    return float(scale) * 1.4426950408889634


def attention_fwd_qscaled(qkv, heads, save_lse=False):
    """Long-sequence 16-bit attention forward (T >= 512, no mask / dropout) on a projection whose q columns already carry
    attention_qscale(scale): qkv [B,T,3*H*64] -> ctx [B,T,H*64] (include/simseg_hip.h: simseg_attention_fwd_qscaled)."""
    require_gpu(qkv)
    B, T, W = qkv.shape
    if W != 3 * heads * 64 or qkv.dtype not in HALF_TYPES or T < ATTN_QSCALED_MIN_T:
        raise ValueError("attention_fwd_qscaled: 16-bit qkv [B, T >= 512, 3*heads*64]")
    out = torch.empty(B, T, heads * 64, device=qkv.device, dtype=qkv.dtype)
    lse = torch.empty(B, heads, T, device=qkv.device, dtype=torch.float32) if save_lse else None
    _push_variant("attention")
    call("simseg_attention_fwd_qscaled", ptr(_c(qkv)), ptr(out), ptr(lse), B, T, heads, stream())
    return out, lse


def attention_fwd_x3(qkv32, heads, scale=0.125):
    """Exact-mode attention forward on the bf16 matrix pipe: qkv32 fp32 [B, T, 3*H*64] -> ctx fp32 [B, T, H*64], every product formed from
    the three exact bf16 pieces of its fp32 operands (six leading piece products, fp32 accumulation: include/simseg_hip.h).  Evaluation
    only (no mask, dropout or saved log-sum-exp)."""
    require_gpu(qkv32)
    B, T, W = qkv32.shape
    if qkv32.dtype != torch.float32 or W != 3 * heads * 64:
        raise ValueError("attention_fwd_x3: fp32 [B, T, 3*H*64]")
    planes = split_bf16x3(_c(qkv32).view(B * T, W), planes=True)           # [3, B*T, W]
    out = torch.empty(B, T, heads * 64, device=qkv32.device, dtype=torch.float32)
    call("simseg_attention_fwd_x3", ptr(planes), B * T * W, ptr(out), B, T, heads, float(scale), stream())
    return out


def split_bf16x3(x2d, b_pattern=False, planes=False):
    """fp32 [rows, K] -> bf16 [rows, 6 K]: the hi / mid / lo bf16 pieces of every element along K, in the A-operand order (hi hi hi mid mid
    lo) or the B-operand order (hi mid lo hi mid hi): gemm(split(a), split(b, True)) is the fp32 product a @ b^T formed on the bf16 MFMA
    pipe in fp32 accumulators (include/simseg_hip.h: simseg_split_bf16x3)."""
    require_gpu(x2d)
    rows, K = x2d.shape
    if x2d.stride(1) != 1:
        raise ValueError("split_bf16x3 needs unit-stride rows")
    # planes: the three pieces as [3, rows, K] (operand form of attention_fwd_x3) instead of the six K-segments of the GEMM operands
    out = torch.empty((3, rows, K) if planes else (rows, 6 * K), device=x2d.device, dtype=torch.bfloat16)
    call("simseg_split_bf16x3", ptr(x2d), ptr(out), rows, K, x2d.stride(0), 2 if planes else int(bool(b_pattern)), stream())
    return out


def transpose_f32(x2d):
    require_gpu(x2d)
    R, C = x2d.shape
    out = torch.empty(C, R, device=x2d.device, dtype=torch.float32)
    call("simseg_transpose_f32", ptr(_c(x2d)), ptr(out), R, C, stream())
    return out


def ragged_maps(mask, multiple, cap, expect=-1):
    """(idx int32 [cap], inv int32 [B*L], row_start int32 [B+1], info int32 [4]) of a [B, L] int64 0/1 mask - include/simseg_hip.h."""
    require_gpu(mask)
    if mask.dtype != torch.int64 or mask.dim() != 2:
        raise TypeError("ragged_maps: int64 [B, L] mask")
    B, L = mask.shape
    dev = mask.device
    idx = torch.empty(cap, device=dev, dtype=torch.int32)
    inv = torch.empty(B * L, device=dev, dtype=torch.int32)
    row_start = torch.empty(B + 1, device=dev, dtype=torch.int32)
    info = torch.empty(4, device=dev, dtype=torch.int32)
    call("simseg_ragged_maps", ptr(_c(mask)), B, L, int(multiple), int(cap), int(expect), ptr(idx), ptr(inv), ptr(row_start), ptr(info), stream())
    return idx, inv, row_start, info


def gather_rows(src2d, idx, out=None):
    """out[i] = src2d[idx[i]] (zero row where idx[i] < 0).  idx int32 on the device."""
    require_gpu(src2d, idx)
    if idx.dtype != torch.int32:
        raise TypeError("gather_rows: idx must be int32")
    n, w = idx.numel(), src2d.shape[-1]
    if out is None:
        out = torch.empty(n, w, device=src2d.device, dtype=src2d.dtype)
    call("simseg_gather_rows", ptr(_c(src2d)), ptr(_c(idx)), ptr(_c(out)), n, w * src2d.element_size(), stream())
    return out


def dropout_apply_(g, seed, p):
    call("simseg_dropout_apply", ptr(_c(g)), dt(g), g.numel(), int(seed), float(p), stream())
    return g


def nce_rows(sims, temperature, target0, ignore_mask=None, smoothing=0.0, write_grad=True):
    """In place: sims <- dLoss/dsims.  Returns out3 = [loss, acc, dLoss/dT] (device tensor)."""
    require_gpu(sims)
    N1, N2 = sims.shape
    scratch = torch.empty(3, N1, device=sims.device, dtype=torch.float32)
    out3 = torch.empty(3, device=sims.device, dtype=torch.float32)
    call("simseg_nce_rows", ptr(_c(sims)), ptr(temperature), ptr(_c(ignore_mask)), ptr(scratch[0]), ptr(scratch[1]), ptr(scratch[2]),
         ptr(out3), N1, N2, int(target0), float(smoothing), int(write_grad), stream())
    return out3


def nce_pair(sims2, temperature, target0, smoothing=0.0, write_grad=True):
    """Both directions at once (include/simseg_hip.h simseg_nce_pair): sims2 [2, N1, N2] fp32, in place -> gradients w.r.t. the
    similarities.  Returns out4 = [loss, i2t acc, t2i acc, dLoss/dT] (device tensor)."""
    require_gpu(sims2)
    _, N1, N2 = sims2.shape
    scratch = torch.empty(6 * N1, device=sims2.device, dtype=torch.float32)
    out4 = torch.empty(4, device=sims2.device, dtype=torch.float32)
    call("simseg_nce_pair", ptr(_c(sims2)), ptr(temperature), ptr(scratch), ptr(out4), N1, N2, int(target0), float(smoothing), int(write_grad), stream())
    return out4


def transpose_multi(mats, scale_flags, scalar=None, alpha=1.0, x0=None):
    """Transposes of up to six contiguous fp32 matrices in ONE launch; matrices whose flag is set are multiplied by alpha * scalar[0] in
    the copy and in place.  Returns (list of transposed tensors, scalar[0] * x0[0] as a 1-element tensor or None)."""
    import ctypes
    n = len(mats)
    for m in mats:
        require_gpu(m)
        if m.dtype != torch.float32 or m.dim() != 2:
            raise TypeError("transpose_multi: contiguous fp32 matrices")
        _c(m)
    outs = [torch.empty(m.shape[1], m.shape[0], device=m.device, dtype=torch.float32) for m in mats]
    y0 = torch.empty(1, device=mats[0].device, dtype=torch.float32) if x0 is not None else None
    PT, I64, I32 = ctypes.c_void_p * n, ctypes.c_int64 * n, ctypes.c_int32 * n
    call("simseg_transpose_multi", PT(*[m.data_ptr() for m in mats]), PT(*[o.data_ptr() for o in outs]), I64(*[m.shape[0] for m in mats]),
         I64(*[m.shape[1] for m in mats]), I32(*[int(bool(f)) for f in scale_flags]), n, ptr(scalar), float(alpha), ptr(x0), ptr(y0), stream())
    return outs, y0


def scale_rows(x, s=None, one_minus=False, alpha=1.0, out=None):
    require_gpu(x)
    if out is None:
        out = torch.empty_like(x)
    D = x.shape[-1]
    call("simseg_scale_rows", ptr(_c(x)), ptr(_c(s)), ptr(out), x.numel() // D, D, int(one_minus), float(alpha), stream())
    return out


def scale_by_scalar(x, scalar, alpha=1.0, out=None):
    if out is None:
        out = torch.empty_like(x)
    call("simseg_scale_by_scalar", ptr(_c(x)), ptr(scalar), ptr(out), x.numel(), float(alpha), stream())
    return out


def retrieval_rank(sim, left_gid, right_gid):
    require_gpu(sim)
    M, N = sim.shape
    has = torch.empty(M, device=sim.device, dtype=torch.int32)
    rank = torch.empty(M, device=sim.device, dtype=torch.int32)
    call("simseg_retrieval_rank", ptr(_c(sim)), ptr(_c(left_gid)), ptr(_c(right_gid)), ptr(has), ptr(rank), M, N, N, stream())
    return has, rank


def retrieval_rank_cols(sim, row_gid, col_gid):
    """Column j of sim [M,N] retrieves rows: (has [N], rank [N]) from the same matrix the row direction used."""
    require_gpu(sim)
    M, N = sim.shape
    has = torch.empty(N, device=sim.device, dtype=torch.int32)
    rank = torch.empty(N, device=sim.device, dtype=torch.int32)
    scratch = torch.empty(N, device=sim.device, dtype=torch.int32)
    call("simseg_retrieval_rank_cols", ptr(_c(sim)), ptr(_c(row_gid)), ptr(_c(col_gid)), ptr(has), ptr(rank), ptr(scratch), M, N, N, stream())
    return has, rank


def recall_counts(has, rank, bounds=(1, 5, 10)):
    counts = torch.empty(4, device=has.device, dtype=torch.int32)
    call("simseg_recall_counts", ptr(has), ptr(rank), has.numel(), int(bounds[0]), int(bounds[1]), int(bounds[2]), ptr(counts), stream())
    return counts


def adamw_step(p, g, m, v, p16, lr, betas, eps, weight_decay, step, grad_scale=1.0):
    call("simseg_adamw_step", ptr(p), ptr(g), ptr(m), ptr(v), ptr(p16), p.numel(), float(lr), float(betas[0]), float(betas[1]),
         float(eps), float(weight_decay), int(step), float(grad_scale), stream())


def tr16_probe():
    out = torch.empty(256, device="cuda", dtype=torch.int32)
    call("simseg_debug_tr16_probe", ptr(out), stream())
    return out.cpu().view(64, 4)


# ---- zero-shot segmentation post-processing (tools/seg_evaluation.py:112-170) ------------------------------------------------
def seg_select(scores, top_cls_num, ncand=5):
    """scores [B,C] fp32 -> (cand_idx [B,ncand] int32 (-1 = slot skipped), cand_score [B,ncand] fp32, threshold [B] fp32)."""
    require_gpu(scores)
    B, C = scores.shape
    idx = torch.empty(B, ncand, device=scores.device, dtype=torch.int32)
    sc = torch.empty(B, ncand, device=scores.device, dtype=torch.float32)
    thr = torch.empty(B, device=scores.device, dtype=torch.float32)
    call("simseg_seg_select", ptr(_c(scores.float())), ptr(idx), ptr(sc), ptr(thr), B, C, int(top_cls_num), int(ncand), stream())
    return idx, sc, thr


def seg_masks(sim, cand_idx, n, want_prob=False):
    """sim [B,nh*nw,C] fp32, cand_idx [B,ncand] -> mask [B,ncand,16nh,16nw] uint8 (0/255; skipped slots all zero), prob or None.
    n: the patch grid - an int (n x n, one resized image) or (nh, nw) (a stitched sliding-window map)."""
    require_gpu(sim, cand_idx)
    B, N, C = sim.shape
    nh, nw = (n, n) if isinstance(n, int) else (int(n[0]), int(n[1]))
    if N != nh * nw:
        raise ValueError(f"seg_masks: sim has {N} patches, expected {nh}x{nw}")
    ncand = cand_idx.shape[1]
    mask = torch.zeros(B, ncand, 16 * nh, 16 * nw, device=sim.device, dtype=torch.uint8)
    prob = torch.zeros(B, ncand, N, device=sim.device, dtype=torch.float32) if want_prob else None
    call("simseg_seg_masks_rect", ptr(_c(sim)), ptr(_c(cand_idx)), ptr(prob) if want_prob else None, ptr(mask), B, nh, nw, C, ncand, stream())
    return mask, prob


def stitch_windows(win, wy, wx, n, step):
    """win [B*wy*wx, n*n, C] fp32 (per-window maps, windows of an image consecutive, row-major over the window grid) -> [B, nh*nw, C]:
    the overlap-average on the source image's patch grid, nh = n + (wy-1)*step, nw = n + (wx-1)*step (simseg_stitch_windows)."""
    require_gpu(win)
    M, N, C = win.shape
    if win.dtype != torch.float32 or N != n * n or M % (wy * wx):
        raise ValueError(f"stitch_windows: fp32 [B*{wy}*{wx}, {n}*{n}, C] maps expected, got {tuple(win.shape)} {win.dtype}")
    B = M // (wy * wx)
    nh, nw = n + (wy - 1) * step, n + (wx - 1) * step
    out = torch.empty(B, nh * nw, C, device=win.device, dtype=torch.float32)
    call("simseg_stitch_windows", ptr(_c(win)), ptr(out), B, wy, wx, n, step, C, stream())
    return out


def morph7(img, erode):
    """7x7 dilate (erode=False) / erode (True), one iteration, on uint8 [..., H, W]."""
    require_gpu(img)
    if img.dtype != torch.uint8:
        raise TypeError("morph7: uint8 images")
    x = _c(img)
    H, W = x.shape[-2:]
    out = torch.empty_like(x)
    call("simseg_morph7", ptr(x), ptr(out), x.numel() // (H * W), H, W, int(bool(erode)), stream())
    return out


def close7(img, valid=None):
    """cv2.dilate then cv2.erode (7x7, one iteration each) in one pass on uint8 [..., H, W]; `valid` (int32, one per image): images
    with a negative entry are skipped and come back zero."""
    require_gpu(img)
    if img.dtype != torch.uint8:
        raise TypeError("close7: uint8 images")
    x = _c(img)
    H, W = x.shape[-2:]
    out = torch.zeros_like(x) if valid is not None else torch.empty_like(x)
    call("simseg_close7", ptr(x), ptr(out), ptr(_c(valid)) if valid is not None else None, x.numel() // (H * W), H, W, stream())
    return out


def seg_predict(masks, cand_idx, cand_score, labels, num_classes, ignore_index=255, hist=None, want_pred=True):
    """masks [B,ncand,Hm,Wm] uint8, labels [B,H,W] uint8 -> (pred [B,H,W] int32 or None, hist [3,C] int64 accumulated:
    rows = intersect, pred area, label area)."""
    require_gpu(masks, labels)
    B, ncand, Hm, Wm = masks.shape
    _, H, W = labels.shape
    if hist is None:
        hist = torch.zeros(3, num_classes, device=masks.device, dtype=torch.int64)
    pred = torch.empty(B, H, W, device=masks.device, dtype=torch.int32) if want_pred else None
    call("simseg_seg_predict", ptr(_c(masks)), ptr(_c(cand_idx)), ptr(_c(cand_score)), ptr(_c(labels)), ptr(pred) if want_pred else None,
         ptr(hist), B, ncand, Hm, Wm, H, W, int(num_classes), int(ignore_index), stream())
    return pred, hist


_CRF_WS = {}
_CRF_SPATIAL = {}      # (device, stream) -> (workspace address, H, W, sxy_g) of the spatial lattice that workspace holds


def dense_crf(rgb, prob, sxy_g=3.0, compat_g=3.0, sxy_b=40.0, srgb=13.0, compat_b=10.0, iters=3, want_q=False):
    """tools/seg_evaluation.py:31-54 for the C candidate maps of each image of a batch.  rgb [B,H,W,3] (or [H,W,3]) uint8 (RGB), prob
    [B,C,H,W] (or [C,H,W]) fp32 in [0,1] -> (mask uint8 0/255, Q(label 1) fp32 or None), shaped like prob."""
    require_gpu(rgb, prob)
    if rgb.dtype != torch.uint8 or prob.dtype != torch.float32:
        raise TypeError("dense_crf: rgb uint8 [B,H,W,3], prob fp32 [B,C,H,W]")
    single = prob.dim() == 3
    if single:
        rgb, prob = rgb[None], prob[None]
    B, C, H, W = prob.shape
    if tuple(rgb.shape) != (B, H, W, 3):
        raise ValueError(f"dense_crf: images {tuple(rgb.shape)} do not match the maps {tuple(prob.shape)}")
    nbytes = raw("simseg_dense_crf_workspace_bytes", B, H, W, C)
    if nbytes < 0:
        raise ValueError(f"dense_crf: 1..8 candidate maps per image (got {C})")
    key = (rgb.device, torch.cuda.current_stream().cuda_stream)
    ws = _CRF_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        _CRF_WS.pop(key, None)
        _CRF_SPATIAL.pop(key, None)
        ws = _CRF_WS[key] = torch.empty(nbytes, device=rgb.device, dtype=torch.uint8)
    mask = torch.empty(B, C, H, W, device=rgb.device, dtype=torch.uint8)
    q = torch.empty(B, C, H, W, device=rgb.device, dtype=torch.float32) if want_q else None
    # the spatial (Gaussian) lattice depends on H, W and sxy only and lives at the head of the workspace: a call with the same three on the
    # same workspace (this module owns it - nobody else writes there) finds it built
    sig = (ws.data_ptr(), int(H), int(W), float(sxy_g))
    reuse = _CRF_SPATIAL.get(key) == sig and os.environ.get("SIMSEG_CRF_SPATIAL_CACHE", "1") != "0"
    call("simseg_dense_crf", ptr(_c(rgb)), ptr(_c(prob)), ptr(mask), ptr(q), B, C, H, W, float(sxy_g), float(compat_g), float(sxy_b), float(srgb),
         float(compat_b), int(iters), ptr(ws), nbytes, int(reuse), stream())
    _CRF_SPATIAL[key] = sig
    if single:
        return mask[0], (q[0] if want_q else None)
    return mask, q
