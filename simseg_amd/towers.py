"""Encoder towers on the HIP kernels: ViT (timm semantics) and BERT (HF semantics), forward and backward in exact fp32 (the
reference without AMP: simseg/core/config.py:50, simseg/core/hooks/optimizer.py:76-77) or bf16 (fp32 accumulate, fp32 master
weights / gradients, fp32 residual stream: the AMP mode the headline runs in).

Each transformer block is ONE torch.autograd.Function whose backward is written out by hand over the C-ABI ops, so
that (a) no T x T / intermediate autograd graph exists, (b) parameter gradients of a block are delivered as soon as
that block's backward finishes -- torch DDP's bucketed RCCL all-reduce overlaps the remaining backward.

Reference arithmetic: timm 0.6.13 VisionTransformer as driven by simseg/models/backbones/mml/vit_builder.py:13-21,
HF BertModel as driven by huggingface_builder.py:16-17 (both third-party; see oracle/simseg_ref.py)."""
import contextlib
import os
import weakref

import torch
from torch.autograd import Function

from . import ops

BF16, F32 = torch.bfloat16, torch.float32


_W16 = {}      # id(parameter) -> (weakref(parameter), parameter._version, parameter.data_ptr(), bf16 copy)


def register_w16(param, w16):
    """The optimizer's kernel has just written `w16` = bf16(param) (raw-pointer update: param._version does not move)."""
    _W16[id(param)] = (weakref.ref(param), param._version, param.data_ptr(), w16)


def drop_split_copy(param):
    """Forget the split-bf16 (exact-mode evaluation) copy of a parameter the optimizer's kernel has just rewritten through raw pointers
    (neither `_version` nor `data_ptr` moves, so `_wt_split`'s own check cannot see the update)."""
    _W3.pop(id(param), None)


def invalidate_weight_cache():
    """Drop every cached bf16 weight copy.  Needed only after writes torch cannot see (in-place ops on `param.data`, which
    carry their own version counter); load_state_dict / copy_ / optimizers / .to() are detected without it."""
    _W16.clear()
    _CAT.clear()
    _W3.clear()


def _wt(w, adt):
    """Weight in the activation dtype (bf16 compute copy of the fp32 master, like autocast's per-step cast).  The copy is
    cached per parameter object and reused while the parameter is untouched: torch-side writes bump `_version`, storage
    swaps change `data_ptr`, and simseg_amd.optim.AdamW refreshes the copy in its own kernel (register_w16)."""
    if adt == F32:
        return w.detach()
    ent = _W16.get(id(w))
    if ent is not None and ent[0]() is w and ent[1] == w._version and ent[2] == w.data_ptr() and ent[3].shape == w.shape and ent[3].dtype == adt:
        return ent[3]
    w16 = ops.cast(w.detach().contiguous(), adt)
    if isinstance(w, torch.nn.Parameter):
        if len(_W16) > 4096:
            for k in [k for k, e in _W16.items() if e[0]() is None]:
                del _W16[k]
        _W16[id(w)] = (weakref.ref(w), w._version, w.data_ptr(), w16)
    return w16


# ---- exact-mode forward GEMMs on the bf16 matrix pipe ---------------------------------------------------------------------------
# gfx950 multiplies bf16 sixteen times faster than fp32 (v_mfma_f32_32x32x16_bf16 vs v_mfma_f32_32x32x2_f32, both accumulating in fp32).
# An fp32 value is exactly the sum of three bf16 pieces, so the fp32 product of the evaluation tools' nn.Linear layers is formed by ONE
# bf16 GEMM over the six leading piece products laid out along K (ops.split_bf16x3): six times the MFMA work at sixteen times the rate,
# with the error of the three dropped products (<= 2^-24 |a||b|) at the level of the fp32 FMA chain's own rounding.  Used for forward
# passes that save nothing for a backward (the evaluation tools; training in exact mode keeps the fp32 kernels, whose operands the
# backward re-reads) and only where the 256x256 bf16 kernel has at least a round of tiles to work on.
_SPLIT_FP32 = os.environ.get("SIMSEG_AMD_SPLIT_FP32", "auto")      # 0 = never, 1 = wherever the shapes allow (tests), auto
_W3 = {}       # id(parameter) -> (weakref(parameter), parameter._version, parameter.data_ptr(), split copy [out, 6 in])
SPLIT_CALLS = [0]


def _split_ok(M, N, K):
    if _SPLIT_FP32 == "0" or K % 64 or N % 8 or M < 256 or N < 128:
        return False
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    return _SPLIT_FP32 == "1" or tiles >= 192


def _wt_split(w):
    ent = _W3.get(id(w))
    if ent is not None and ent[0]() is w and ent[1] == w._version and ent[2] == w.data_ptr() and ent[3].shape[0] == w.shape[0]:
        return ent[3]
    w3 = ops.split_bf16x3(w.detach().reshape(w.shape[0], -1).contiguous(), b_pattern=True)
    if isinstance(w, torch.nn.Parameter):
        if len(_W3) > 4096:
            for k in [k for k, e in _W3.items() if e[0]() is None]:
                del _W3[k]
        _W3[id(w)] = (weakref.ref(w), w._version, w.data_ptr(), w3)
    return w3


_WQS = {}


def _wt_qscaled(w, b, adt, D):
    """timm Attention.qkv for the long-sequence 16-bit forward (ops.attention_fwd_qscaled): the 16-bit copy of the weight [3D, D] and the
    fp32 bias [3D] with the q rows multiplied by softmax scale * log2(e) in fp32, BEFORE the rounding to 16 bits - the projection then
    delivers q * scale * log2(e) rounded once, and the attention kernel's exponent is its MFMA output.  Cached per parameter like _wt."""
    key = (id(w), id(b))
    ent = _WQS.get(key)
    if (ent is not None and ent[0]() is w and ent[1]() is b and ent[2] == (w._version, b._version, w.data_ptr(), b.data_ptr())
            and ent[3].dtype == adt):
        return ent[3], ent[4]
    c = ops.attention_qscale(64 ** -0.5)
    ws = w.detach().float().clone()
    ws[:D] *= c
    bs = b.detach().float().clone()
    bs[:D] *= c
    w16 = ops.cast(ws.contiguous(), adt)
    if isinstance(w, torch.nn.Parameter):
        if len(_WQS) > 1024:
            for k in [k for k, e in _WQS.items() if e[0]() is None]:
                del _WQS[k]
        _WQS[key] = (weakref.ref(w), weakref.ref(b), (w._version, b._version, w.data_ptr(), b.data_ptr()), w16, bs)
    return w16, bs


def _fwd_gemm(x2d, w, adt, saving, **kw):
    """x2d @ w^T for an nn.Linear weight `w` [out, in] in the forward pass: the activation-dtype kernel, or - exact mode, nothing saved for
    a backward, enough tiles - the split-bf16 form of the same fp32 product."""
    if adt == F32 and not saving and x2d.dtype == F32 and _split_ok(x2d.shape[0], w.shape[0], x2d.shape[1]):
        SPLIT_CALLS[0] += 1
        kw.setdefault("out_dtype", F32)
        return ops.gemm(ops.split_bf16x3(x2d), _wt_split(w), **kw)
    return ops.gemm(x2d, _wt(w, adt) if w.dim() == 2 else _wt(w.reshape(w.shape[0], -1), adt), **kw)


import threading

_GRAD_MODE = threading.local()


class _GradAwareFn(Function):
    """autograd switches grad mode OFF inside Function.forward, and ctx.needs_input_grad reports the inputs' requires_grad flags whatever
    the caller's mode: a forward cannot tell torch.no_grad() evaluation from training by itself.  The caller's mode is recorded here, on
    the way in."""

    @classmethod
    def apply(cls, *args):
        _GRAD_MODE.on = torch.is_grad_enabled()
        return super().apply(*args)


def _saving(ctx):
    """Does this forward need to keep anything for a backward?  Under torch.no_grad() - every evaluation tool - the parameters still
    'need' gradients as far as ctx.needs_input_grad is concerned, and the blocks used to save statistics, log-sum-exps and GELU
    derivatives nobody would read."""
    if os.environ.get("SIMSEG_AMD_EVAL_SAVES") == "1":      # A/B runs: the round-2 behaviour (evaluation forwards save as training ones do)
        return any(ctx.needs_input_grad)
    return getattr(_GRAD_MODE, "on", True) and any(ctx.needs_input_grad)


def _as_one(*ts):
    """The tensors as ONE tensor (rows stacked) when they already lie back to back in the same storage, else None."""
    a = ts[0]
    base, off = a.untyped_storage().data_ptr(), a.storage_offset()
    for t in ts:
        if (t.dtype != a.dtype or t.shape[1:] != a.shape[1:] or not t.is_contiguous() or t.untyped_storage().data_ptr() != base
                or t.storage_offset() != off):
            return None
        off += t.numel()
    return torch.as_strided(a, (sum(t.shape[0] for t in ts),) + tuple(a.shape[1:]), a.stride(), a.storage_offset())


_CAT = {}      # id(first parameter) -> (the _W16 entries the concatenation was made from, concatenated bf16 weight)


def _wt_stacked(ws, adt):
    """[sum out, in] weight of several nn.Linear modules that share their input (HF keeps BERT's query / key / value as three
    modules; one GEMM computes all three).  No copy when the operands are adjacent in memory: the fp32 masters are packed by the module
    (nn._BertSelf), the bf16 copies by the optimizer (optim.AdamW lays matrices out in parameter order).  Otherwise (bf16 without our
    optimizer: evaluation) the concatenation is made once and reused until one of the bf16 copies is refreshed."""
    parts = [_wt(w, adt) for w in ws]
    one = _as_one(*parts)
    if one is not None:
        return one
    if adt != F32:
        ents = [_W16.get(id(w)) for w in ws]
        hit = _CAT.get(id(ws[0]))
        if hit is not None and all(e is not None and e is h for e, h in zip(ents, hit[0])):
            return hit[1]
        cat = torch.cat(parts)
        if all(e is not None for e in ents):
            _CAT[id(ws[0])] = (ents, cat)
        return cat
    return torch.cat(parts)


def _splitk(m, n, k_rows):
    tiles = ((m + 127) // 128) * ((n + 127) // 128)
    nk = (k_rows + 63) // 64
    return max(1, min(nk, (1024 + tiles - 1) // tiles, 64))


def _zeros(device, *shapes):
    """fp32 zero tensors for one block's accumulated gradients, carved from ONE zero-filled buffer (one fill launch per block
    instead of one per tensor; each view starts on a 256-byte boundary)."""
    sizes = [int(torch.Size(sh).numel()) if not isinstance(sh, int) else sh for sh in shapes]
    offs, o = [], 0
    for n in sizes:
        offs.append(o)
        o += (n + 63) // 64 * 64
    buf = torch.zeros(o, device=device, dtype=F32)
    return [buf[a:a + n].view(sh) for a, n, sh in zip(offs, sizes, shapes)]


# The saved GELU' of the MLP blocks as an 8-bit image (simseg_gemm act 7 / 8) instead of a 16-bit one (5 / 6): half of the step's largest
# epilogue stream (1.24 GB per ViT-B layer written + read) at the same accuracy of the product it feeds - relative RMS error 2.7e-3 against
# 2.5e-3, the product's own rounding to 16 bits being 1.7e-3 (tests/test_gpu_kernels.py).  SIMSEG_AMD_GELU_GRAD_BITS=16: the
# 16-bit image.
_GELU8 = os.environ.get("SIMSEG_AMD_GELU_GRAD_BITS", "8") != "16"
# The gradient of the ViT residual stream between the LayerNorm backward kernels of the 16-bit training modes: handed on as the 16-bit copy
# each kernel writes anyway (every GEMM that consumes it reads 16-bit operands), no fp32 image written or read - 0.93 GB per ViT-B layer and
# step less.  Measured on every parameter gradient against the exact-fp32 backward (ViT-B + BERT-base, B = 256): mean 1 - cosine 5.06e-4
# against 5.04e-4 with the fp32 stream (tests/test_gpu_fullsize.py).  The fp32 tensor autograd passes from block to block is then an
# UNWRITTEN allocation marked `_simseg_lazy32`; only this module's functions produce and consume it (vit_forward's chain), and a consumer
# that finds the mark without a valid 16-bit shadow refuses loudly.  SIMSEG_AMD_RESGRAD_BITS=32: the fp32 stream.
_RES16 = os.environ.get("SIMSEG_AMD_RESGRAD_BITS", "16") != "32"
# LayerNorm backward of the ViT blocks in the 16-bit modes: the normalised input from the layer's saved 16-bit output (simseg_layernorm_bwd
# y_bf16) instead of from the fp32 input - 0 = from x.  Same fidelity harness: mean 1 - cosine 4.642e-4 vs 4.638e-4 with gains spread over
# a decade and offsets of their order.
_XHAT_Y = os.environ.get("SIMSEG_AMD_LN_BWD_FROM_OUTPUT", "1") != "0"


def _lazy32(shape, device, dx16, dsum):
    """The fp32 gradient tensor autograd wants, NOT written: its values live in the 16-bit shadow only."""
    dx = torch.empty(shape, device=device, dtype=F32)
    dx._simseg_lazy32 = True
    _put_shadow(dx, dx16.view(shape), dsum)
    return dx


def _need_shadow(t32, sh):
    if sh is None and getattr(t32, "_simseg_lazy32", False):
        raise RuntimeError("a residual-stream gradient that exists only as its 16-bit shadow (SIMSEG_AMD_RESGRAD_BITS=16) reached a consumer without a "
                           "valid shadow - was it modified in place, or consumed twice?  Set SIMSEG_AMD_RESGRAD_BITS=32")


def _grad_target(p):
    """Where a data-parallel gradient exchange wants parameter p's gradient written: a fresh, zero-filled view of its flat buffer
    (simseg_amd/parallel.py GradSync.begin()) - the split-K weight-gradient GEMMs accumulate into a zeroed output anyway, and autograd adopts
    the returned view as `.grad`, so the exchange has nothing to copy - or None (no exchange / not armed: a private zero buffer)."""
    f = getattr(p, "_simseg_grad_target", None)
    return f() if f is not None else None


def _zeros_or(device, targets, *shapes):
    """_zeros(), except that entry i is `targets[i]` where that is a tensor of the right shape."""
    use = [t is not None and tuple(t.shape) == tuple(torch.Size(sh if not isinstance(sh, int) else (sh,))) for t, sh in zip(targets, shapes)]
    rest = iter(_zeros(device, *[sh for sh, u in zip(shapes, use) if not u]))
    return [t if u else next(rest) for t, u in zip(targets, use)]


# Weight gradients on their own stream (opt-in experiment, SIMSEG_AMD_WGRAD_STREAM=1): a block's four weight-gradient GEMMs depend only on
# tensors its backward already has and nothing reads their results before the optimizer, so they can run beside the data-gradient chain
# (dgrad -> LayerNorm backward -> attention backward ...) instead of inside it.  One extra stream per stream the backward runs on (the two
# towers have their own); joined before the block's backward returns.
_WG_ON = os.environ.get("SIMSEG_AMD_WGRAD_STREAM", "0") == "1"
_WG_STREAMS = {}
_WG_PENDING = threading.local()


def _wg_side(cur):
    st = _WG_STREAMS.get(cur.cuda_stream)
    if st is None:
        st = _WG_STREAMS[cur.cuda_stream] = torch.cuda.Stream(device=cur.device)
    return st


def _wg_join():
    pend = getattr(_WG_PENDING, "streams", None)
    if pend:
        cur = torch.cuda.current_stream()
        for st in pend:
            cur.wait_stream(st)
        pend.clear()


def _joins_wgrad(fn):
    def wrapped(ctx, *grads):
        try:
            return fn(ctx, *grads)
        finally:
            _wg_join()
    return wrapped


def _wgrad(dy, x, out=None):
    if _WG_ON and dy.is_cuda:
        cur = torch.cuda.current_stream()
        side = _wg_side(cur)
        if out is None:
            out = torch.zeros(dy.shape[1], x.shape[1], device=dy.device, dtype=F32)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            _wgrad_now(dy, x, out)
        pend = getattr(_WG_PENDING, "streams", None)
        if pend is None:
            pend = _WG_PENDING.streams = []
        if side not in pend:
            pend.append(side)
        return out
    return _wgrad_now(dy, x, out)


def _wgrad_now(dy, x, out=None):
    """dW[out,in] = dy^T . x  (contraction over rows), fp32, split-K atomics into the zero-filled `out`.  bf16 operands go to the
    transposed-operand MFMA kernel; fp32 operands (exact mode) are transposed first - the fp32 kernel is row.row only."""
    out_f, in_f = dy.shape[1], x.shape[1]
    dw = torch.zeros(out_f, in_f, device=dy.device, dtype=F32) if out is None else out
    sk = _splitk(out_f, in_f, dy.shape[0])
    if dy.dtype == F32:
        ops.gemm(ops.transpose_f32(dy), ops.transpose_f32(x), out=dw, accumulate=True, splitk=sk)
    else:
        ops.gemm(dy, x, trans_a=True, trans_b=True, out=dw, accumulate=True, splitk=sk)
    return dw


def _dgrad(d, w_, **kw):
    """dx = d . W for a weight stored [out, in] (nn.Linear layout), with the GEMM epilogue options in kw."""
    if d.dtype == F32:
        return ops.gemm(d, ops.transpose_f32(w_), **kw)
    return ops.gemm(d, w_, trans_b=True, **kw)


_FUSED_QKV_BIAS = os.environ.get("SIMSEG_AMD_FUSED_QKV_BIAS", "1") != "0"      # (A/B runs: 0 = a column-sum pass over dqkv instead)


def _zeros_like_bias(out, n, device):
    """The accumulation target of a fused bias gradient: the caller's zeroed buffer, or a fresh one."""
    return torch.zeros(n, device=device, dtype=F32) if out is None else out


def _bgrad(dy, out=None):
    db = torch.zeros(dy.shape[1], device=dy.device, dtype=F32) if out is None else out
    ops.colsum_accum(dy, db)
    return db


def _act_grad(t32, adt):
    """A gradient of the fp32 residual stream as a GEMM operand of the compute dtype."""
    return t32 if adt == F32 else ops.cast(t32, adt)


def _ln_bwd(adt, x, mean, rstd, w, dg, db, dy, dres=None, dy32=None, dxsum=None, drop=(0.0, 0), dres16=None, want32=True, y16=None, beta=None):
    """Backward of a LayerNorm whose input was `x`: dx = LN'(dy [+ dy32]) + dres.  Returns (dx fp32, dx in the compute dtype with the
    dropout mask `drop` = (p, seed) of the dense layer that produced x re-applied); column sums of the second go to dxsum (that
    layer's bias gradient).  In bf16 mode all of it is one kernel; in exact mode the pieces are separate launches."""
    if adt != F32:
        # (dres16 / want32=False: the residual-stream gradient arrives as - and leaves only as - the 16-bit copy, see _RES16)
        if not (_XHAT_Y and y16 is not None and beta is not None):
            y16 = beta = None
        return ops.layernorm_bwd(x, mean, rstd, w, dg, db, dy16=dy, dy32=dy32, dres=dres, dres16=dres16, dxsum=dxsum, drop_seed=drop[1], drop_p=drop[0],
                                 want_bf16=adt, want_f32=want32, y16=y16, beta=beta)
    if dy32 is not None:
        raise ValueError("exact mode: fold the second gradient into `dy` with the producing GEMM's residual epilogue")
    dx32, _ = ops.layernorm_bwd(x, mean, rstd, w, dg, db, dy32=dy, dres=dres, want_bf16=False)
    d = dx32
    if drop[0] > 0:
        d = ops.dropout_apply_(dx32.clone(), drop[1], drop[0])
    if dxsum is not None:
        ops.colsum_accum(d.view(-1, d.shape[-1]), dxsum)
    return dx32, d


# bf16 copy + column sums of a gradient tensor handed from one block's backward to the next (saves a cast pass and a column-sum
# pass per block).  The pair rides on the fp32 gradient tensor itself (autograd hands the same tensor object to the consumer node;
# its Python attributes travel with it) together with the tensor's version counter: if autograd accumulated another gradient into
# the buffer in place (a block output with two consumers), or anything else wrote to it, the version has moved and the shadow is
# ignored -- the consumer then casts / sums the tensor it was actually given.  Nothing is keyed by address, nothing is global.
SHADOW_HITS = [0]     # consumed shadows (tests assert that the fast path is the one that runs)


def _put_shadow(t32, t16, colsum):
    t32._simseg_shadow = (t16, colsum, t32._version)


def _take_shadow(t32, adt=None):
    sh = getattr(t32, "_simseg_shadow", None)
    if sh is None:
        return None
    del t32._simseg_shadow
    if sh[2] != t32._version or sh[0].numel() != t32.numel() or (adt is not None and sh[0].dtype != adt):
        return None
    SHADOW_HITS[0] += 1
    return sh


# ------------------------------------------------------------------------------------------------------------------
# generic pieces
# ------------------------------------------------------------------------------------------------------------------
class LinearFn(_GradAwareFn):
    """y = x W^T (+ b) on [..., in] -> [..., out]; fp32 in / fp32 out at the module boundary."""

    @staticmethod
    def forward(ctx, x, w, b, adt):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1]).contiguous()
        xa = x2 if adt == F32 else ops.cast(x2, adt)
        ctx.adt, ctx.has_b, ctx.shp = adt, b is not None, shp
        if not _saving(ctx):      # evaluation: nothing saved (exact mode: the split-bf16 form where the problem is large)
            return _fwd_gemm(xa, w, adt, False, bias=None if b is None else b.detach(), out_dtype=F32).view(*shp[:-1], w.shape[0])
        wa = _wt(w, adt)
        y = ops.gemm(xa, wa, bias=None if b is None else b.detach(), out_dtype=F32)
        ctx.save_for_backward(xa, wa)
        return y.view(*shp[:-1], w.shape[0])

    @staticmethod
    @_joins_wgrad
    def backward(ctx, dy):
        xa, wa = ctx.saved_tensors
        d16 = _act_grad(dy.reshape(-1, dy.shape[-1]).contiguous(), ctx.adt)
        dx = _dgrad(d16, wa, out_dtype=F32).view(ctx.shp) if ctx.needs_input_grad[0] else None
        dw = _wgrad(d16, xa) if ctx.needs_input_grad[1] else None
        db = _bgrad(d16) if (ctx.has_b and ctx.needs_input_grad[2]) else None
        return dx, dw, db, None


class LayerNormFn(Function):
    @staticmethod
    def forward(ctx, x, w, b, eps, adt=None, lazy_ok=False):
        x = x.contiguous()
        # (adt = a 16-bit compute type: the kernel writes a copy of y in it next to the fp32 rows, picked up by the consumer - ProjectPoolFn
        #  through y._simseg_fwd16 - instead of a cast pass over the features)
        y, y16, mean, rstd = ops.layernorm_fwd(x, w.detach(), b.detach(), eps, save_stats=True, want_bf16_copy=adt if adt in ops.HALF_TYPES else False)
        if y16 is not None:
            y._simseg_fwd16 = (y16, y._version)
        ctx.lazy = adt in ops.HALF_TYPES and _RES16 and lazy_ok
        ctx.adt16 = adt if adt in ops.HALF_TYPES else True          # (the shadow's type: the consumer accepts only its own compute type)
        ctx.save_for_backward(x, mean, rstd, w.detach())
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, w = ctx.saved_tensors
        dg, db, dsum = torch.zeros_like(w), torch.zeros_like(w), torch.zeros_like(w)
        dx, dx16 = ops.layernorm_bwd(x, mean, rstd, w, dg, db, dy32=dy.contiguous(), dxsum=dsum, want_f32=not ctx.lazy, want_bf16=ctx.adt16)
        if dx is None:       # (lazy_ok: the producer of x is a ViTBlockFn, which reads the shadow only)
            dx = _lazy32(x.shape, x.device, dx16, dsum)
        else:
            _put_shadow(dx, dx16, dsum)
        return dx, dg, db, None, None, None


# ------------------------------------------------------------------------------------------------------------------
# ViT
# ------------------------------------------------------------------------------------------------------------------
class ViTEmbedFn(_GradAwareFn):
    """patch_embed -> cat(cls) -> + pos_embed  (vit_builder.py:14-17) -> fp32 residual stream [B,1+N,D]."""

    @staticmethod
    def forward(ctx, image, pw, pb, cls, pos, adt):
        B, _, H, W = image.shape
        D = pw.shape[0]
        N = (H // 16) * (W // 16)
        if pos.shape[1] != N + 1:
            raise ValueError(f"pos_embed has {pos.shape[1]} tokens but the {H}x{W} input makes {N + 1}")
        cols = ops.vit_im2col(image.contiguous().float(), adt)
        x = torch.empty(B, N + 1, D, device=image.device, dtype=F32)
        _fwd_gemm(cols, pw, adt, _saving(ctx), bias=pb.detach(), residual=pos.detach().reshape(N + 1, D), row_group=N, res_mod=True,
                  out=x.view(-1, D))
        ops.vit_cls_rows(cls.detach().reshape(-1), pos.detach().reshape(-1), x)
        ctx.adt, ctx.dims = adt, (B, N, D)
        ctx.save_for_backward(cols)
        return x

    @staticmethod
    @_joins_wgrad
    def backward(ctx, dx):
        (cols,) = ctx.saved_tensors
        B, N, D = ctx.dims
        dx = dx.contiguous()
        sh = _take_shadow(dx, ctx.adt) if ctx.adt != F32 else None
        _need_shadow(dx, sh)
        lazy = getattr(dx, "_simseg_lazy32", False)
        if sh is not None:       # the first block's backward left a 16-bit copy of dx: slice THAT (half the bytes of the fp32 slice, no cast pass; same bits)
            dx16 = sh[0].view(B, N + 1, D)
            dp16 = dx16[:, 1:].contiguous().view(-1, D)
        else:
            dp16 = _act_grad(dx[:, 1:].contiguous().view(-1, D), ctx.adt)
        dw = _wgrad(dp16, cols).view(D, 3, 16, 16) if ctx.needs_input_grad[1] else None
        db = _bgrad(dp16) if ctx.needs_input_grad[2] else None
        dcls = dpos = None
        if ctx.needs_input_grad[3]:
            if lazy:             # (the fp32 image of dx was never written: the [cls] rows of its 16-bit shadow)
                dcls = dx16[:, 0].float().sum(0).view(1, 1, D)
            else:
                dcls = torch.zeros(D, device=dx.device, dtype=F32)
                ops.vit_cls_grad(dx, dcls)
                dcls = dcls.view(1, 1, D)
        if ctx.needs_input_grad[4]:
            dpos = torch.zeros((N + 1) * D, device=dx.device, dtype=F32)
            ops.colsum_accum((dx16 if lazy else dx).view(B, (N + 1) * D), dpos)
            dpos = dpos.view(1, N + 1, D)
        return None, dw, db, dcls, dpos, None


class ViTBlockFn(_GradAwareFn):
    """timm Block: x + proj(attn(norm1 x)); then + fc2(gelu(fc1(norm2 .)))   (pre-LN, eps 1e-6, erf GELU)."""

    @staticmethod
    def forward(ctx, x, heads, adt, n1w, n1b, qw, qb, pw, pb, n2w, n2b, f1w, f1b, f2w, f2b, lazy_ok=False):
        # lazy_ok: the producer of x is another function of this module that reads the 16-bit shadow of the gradient returned here and
        # never its fp32 values (vit_forward's chain) - see _RES16
        B, T, D = x.shape
        x = x.contiguous()
        save = _saving(ctx)
        ln1, _, mean1, rstd1 = ops.layernorm_fwd(x, n1w.detach(), n1b.detach(), 1e-6, out_dtype=adt, save_stats=save)
        if not save and adt != F32 and T >= ops.ATTN_QSCALED_MIN_T and os.environ.get("SIMSEG_AMD_QSCALED", "1") != "0":
            # evaluation, 16-bit, long sequences (384^2 / 512^2 windows): the 64-queries-per-wave forward on a projection that already
            # carries the softmax scale (SIMSEG_AMD_QSCALED=0: A/B runs on the kernel that scales q itself)
            qw_s, qb_s = _wt_qscaled(qw, qb, adt, D)
            qkv = ops.gemm(ln1.view(-1, D), qw_s, bias=qb_s)
            att, _ = ops.attention_fwd_qscaled(qkv.view(B, T, 3 * D), heads)
            x1 = _fwd_gemm(att.view(-1, D), pw, adt, False, bias=pb.detach(), residual=x.view(-1, D), out_dtype=F32)
            ln2, _, _, _ = ops.layernorm_fwd(x1, n2w.detach(), n2b.detach(), 1e-6, out_dtype=adt, save_stats=False)
            act = _fwd_gemm(ln2, f1w, adt, False, bias=f1b.detach(), act=1)
            return _fwd_gemm(act, f2w, adt, False, bias=f2b.detach(), residual=x1, out_dtype=F32).view(B, T, D)
        if not save:        # evaluation: no operand is kept (exact mode: large problems go through the split-bf16 form of the fp32 products)
            qkv = _fwd_gemm(ln1.view(-1, D), qw, adt, False, bias=qb.detach())
            if adt == F32 and _SPLIT_FP32 != "0" and (_SPLIT_FP32 == "1" or (T >= 512 and B * heads >= 128)):
                # exact mode, long sequences: the attention products through the bf16 pieces as well (the fp32 MFMA kernel is 20 % of an
                # exact-mode ViT-B@512 evaluation once the GEMMs are split; measured 2.07 -> 1.58 ms per layer at 63 x 1025 tokens with the
                # split pass, a tie at T = 325, slower at T = 197: tools/attn_x3_bench.py)
                SPLIT_CALLS[0] += 1
                att = ops.attention_fwd_x3(qkv.view(B, T, 3 * D), heads, scale=64 ** -0.5)
            else:
                att, _ = ops.attention_fwd(qkv.view(B, T, 3 * D), heads, None, scale=64 ** -0.5, save_lse=False)
            x1 = _fwd_gemm(att.view(-1, D), pw, adt, False, bias=pb.detach(), residual=x.view(-1, D), out_dtype=F32)
            ln2, _, _, _ = ops.layernorm_fwd(x1, n2w.detach(), n2b.detach(), 1e-6, out_dtype=adt, save_stats=False)
            act = _fwd_gemm(ln2, f1w, adt, False, bias=f1b.detach(), act=1)
            return _fwd_gemm(act, f2w, adt, False, bias=f2b.detach(), residual=x1, out_dtype=F32).view(B, T, D)
        qw_, pw_, f1w_, f2w_ = _wt(qw, adt), _wt(pw, adt), _wt(f1w, adt), _wt(f2w, adt)
        qkv = ops.gemm(ln1.view(-1, D), qw_, bias=qb.detach())
        att, lse = ops.attention_fwd(qkv.view(B, T, 3 * D), heads, None, scale=64 ** -0.5, save_lse=save)
        x1 = ops.gemm(att.view(-1, D), pw_, bias=pb.detach(), residual=x.view(-1, D), out_dtype=F32)
        ln2, _, mean2, rstd2 = ops.layernorm_fwd(x1, n2w.detach(), n2b.detach(), 1e-6, out_dtype=adt, save_stats=save)
        # (16-bit modes, full tiles: GELU' is saved as the tile-blocked accumulator image the dgrad through fc2 reads back - act 5 / 6, or
        #  7 / 8 = the same image with one byte per element, _GELU8)
        blk = save and adt != F32 and ops.gemm_aux_blocked_ok(B * T, 4 * D, D)
        g8 = blk and _GELU8
        pre = torch.empty(B * T, 4 * D, device=x.device, dtype=torch.uint8 if g8 else adt) if save else None
        act = ops.gemm(ln2, f1w_, bias=f1b.detach(), act=((7 if g8 else 5) if blk else 3) if save else 1, aux_out=pre)     # pre holds GELU'(fc1 output)
        y = ops.gemm(act, f2w_, bias=f2b.detach(), residual=x1, out_dtype=F32)
        ctx.adt, ctx.heads, ctx.dims, ctx.blk = adt, heads, (B, T, D), (2 if g8 else 1) if blk else 0
        ctx.lazy = bool(lazy_ok) and adt != F32 and _RES16
        ctx.wparams = (f2w, f1w, pw, qw)                                  # (the parameter objects: _grad_target in the backward)
        if save:
            ctx.save_for_backward(x, mean1, rstd1, ln1, qkv, att, lse, x1, mean2, rstd2, ln2, pre, act, qw_, pw_, f1w_, f2w_,
                                  n1w.detach(), n2w.detach(), n1b.detach(), n2b.detach())
        return y.view(B, T, D)

    @staticmethod
    @_joins_wgrad
    def backward(ctx, dy):
        x, mean1, rstd1, ln1, qkv, att, lse, x1, mean2, rstd2, ln2, pre, act, qw_, pw_, f1w_, f2w_, n1w, n2w, n1b, n2b = ctx.saved_tensors
        B, T, D = ctx.dims
        adt = ctx.adt
        need = ctx.needs_input_grad
        dy = dy.contiguous()
        sh = _take_shadow(dy, adt) if adt != F32 else None
        _need_shadow(dy, sh)
        res16 = sh is not None and adt != F32 and _RES16      # the residual gradient is taken from its 16-bit shadow (the fp32 image may not even exist)
        dy = dy.view(-1, D)
        if sh is not None:
            dy16, df2b = sh[0].view(-1, D), sh[1]
        else:
            dy16 = _act_grad(dy, adt)
            df2b = _bgrad(dy16) if need[14] else None
        tg = [_grad_target(w) if nd else None for w, nd in zip(ctx.wparams, (need[13], need[11], need[7], need[5]))]
        (df1b, df2w_z, df1w_z, dn2w, dn2b, dpb, dpw_z, dqw_z, dqb_z, dn1w, dn1b, dsum) = _zeros_or(
            dy.device, (None, tg[0], tg[1], None, None, None, tg[2], tg[3], None, None, None, None),
            (4 * D,), (D, 4 * D), (4 * D, D), (D,), (D,), (D,), (D, D), (3 * D, D), (3 * D,), (D,), (D,), (D,))
        # mlp
        dpre = _dgrad(dy16, f2w_, act=(4, 6, 8)[ctx.blk], aux=pre, colsum=df1b)
        df2w = _wgrad(dy16, act, df2w_z) if need[13] else None
        dln2 = _dgrad(dpre, f1w_)
        df1w = _wgrad(dpre, ln2, df1w_z) if need[11] else None
        r16 = adt != F32 and _RES16                           # between this block's two LayerNorm backward kernels: 16 bits only
        dx1_32, dx1_16 = _ln_bwd(adt, x1, mean2, rstd2, n2w, dn2w, dn2b, dln2, dres=None if res16 else dy, dres16=dy16 if res16 else None, dxsum=dpb,
                                 want32=not r16, y16=ln2, beta=n2b)
        # attention
        datt = _dgrad(dx1_16, pw_)
        dpw = _wgrad(dx1_16, att.view(-1, D), dpw_z) if need[7] else None
        dqb = _zeros_like_bias(dqb_z, 3 * D, qkv.device) if need[6] else None       # the qkv bias gradient rides on the attention backward
        dqkv = ops.attention_bwd(qkv.view(B, T, 3 * D), att, datt.view(B, T, D), lse, ctx.heads, None, scale=64 ** -0.5,
                                 colsum=dqb if _FUSED_QKV_BIAS else None).view(-1, 3 * D)
        if dqb is not None and not _FUSED_QKV_BIAS:
            ops.colsum_accum(dqkv, dqb)
        dln1 = _dgrad(dqkv, qw_)
        dqw = _wgrad(dqkv, ln1.view(-1, D), dqw_z) if need[5] else None
        dx, dx16 = _ln_bwd(adt, x.view(-1, D), mean1, rstd1, n1w, dn1w, dn1b, dln1, dres=None if r16 else dx1_32, dres16=dx1_16 if r16 else None,
                           dxsum=dsum, want32=not ctx.lazy, y16=ln1, beta=n1b)
        if dx is None:
            dx = _lazy32((B, T, D), x.device, dx16, dsum)
        else:
            dx = dx.view(B, T, D)
            if adt != F32:
                _put_shadow(dx, dx16, dsum)
        return (dx, None, None, dn1w, dn1b, dqw, dqb, dpw, dpb, dn2w, dn2b, df1w, df1b, df2w, df2b, None)


def vit_forward(m, image, adt):
    """m: module tree with timm parameter names (see simseg_amd/nn.py ViT). Returns all tokens after the final LN."""
    x = ViTEmbedFn.apply(image, m.patch_embed.proj.weight, m.patch_embed.proj.bias, m.cls_token, m.pos_embed, adt)
    for blk in m.blocks:
        x = ViTBlockFn.apply(x, m.num_heads, adt, blk.norm1.weight, blk.norm1.bias, blk.attn.qkv.weight, blk.attn.qkv.bias,
                             blk.attn.proj.weight, blk.attn.proj.bias, blk.norm2.weight, blk.norm2.bias,
                             blk.mlp.fc1.weight, blk.mlp.fc1.bias, blk.mlp.fc2.weight, blk.mlp.fc2.bias, True)      # (lazy_ok: the chain below)
    return LayerNormFn.apply(x, m.norm.weight, m.norm.bias, 1e-6, adt, True)


# ------------------------------------------------------------------------------------------------------------------
# BERT
# ------------------------------------------------------------------------------------------------------------------
class BertEmbedFn(Function):
    """HF BertEmbeddings: LN(word[ids] + pos[:L] + type[0]) -> dropout."""

    @staticmethod
    def forward(ctx, ids, mask, word, pos, typ, lnw, lnb, drop_p, seed):
        B, L = ids.shape
        if L > pos.shape[0]:
            raise ValueError(f"sequence length {L} exceeds max_position_embeddings {pos.shape[0]}")
        s = ops.bert_embed_fwd(ids.contiguous(), word.detach(), pos.detach(), typ.detach()[0].contiguous())
        y, _, mean, rstd = ops.layernorm_fwd(s, lnw.detach(), lnb.detach(), 1e-12, save_stats=True)
        if drop_p > 0:
            ops.dropout_apply_(y, seed, drop_p)
        ctx.drop = (drop_p, seed)
        ctx.shapes = (word.shape, pos.shape, typ.shape)
        ctx.wparams = (word,)
        ctx.save_for_backward(ids, mask, s, mean, rstd, lnw.detach())
        return y

    @staticmethod
    def backward(ctx, dy):
        ids, mask, s, mean, rstd, lnw = ctx.saved_tensors
        B, L = ids.shape
        D = s.shape[-1]
        dy = dy.contiguous()
        if ctx.drop[0] > 0:
            dy = ops.dropout_apply_(dy.clone(), ctx.drop[1], ctx.drop[0])
        dg, db = torch.zeros_like(lnw), torch.zeros_like(lnw)
        ds, _ = ops.layernorm_bwd(s, mean, rstd, lnw, dg, db, dy32=dy, want_bf16=False)
        wshape, pshape, tshape = ctx.shapes
        dword = dpos = dtyp = None
        if ctx.needs_input_grad[2]:
            dword = _grad_target(ctx.wparams[0])                      # (scatter-add into zeros: the exchange buffer's view will do)
            if dword is None or dword.shape != wshape:
                dword = torch.zeros(wshape, device=dy.device, dtype=F32)
            ops.bert_embed_bwd(ids, mask, ds, dword)
        if ctx.needs_input_grad[3]:
            dpos = torch.zeros(pshape, device=dy.device, dtype=F32)
            ops.colsum_accum(ds.view(B, L * D), dpos.view(-1)[: L * D])
        if ctx.needs_input_grad[4]:
            dtyp = torch.zeros(tshape, device=dy.device, dtype=F32)
            ops.colsum_accum(ds.view(B * L, D), dtyp[0])
        return None, None, dword, dpos, dtyp, dg, db, None, None


class RowMapFn(Function):
    """y[i] = x[fwd_map[i]] (a zero row where the map is -1); the backward is the same gather with the inverse map.  With
    fwd_map = the flat positions of the real tokens it PACKS a ragged caption batch [B*L, D] -> [Nv, D]; with the inverse map (-1 at
    padded positions) it puts the rows back."""

    @staticmethod
    def forward(ctx, x2d, fwd_map, bwd_map):
        ctx.save_for_backward(bwd_map)
        return ops.gather_rows(x2d.contiguous(), fwd_map)

    @staticmethod
    def backward(ctx, dy):
        (bwd_map,) = ctx.saved_tensors
        return ops.gather_rows(dy.contiguous(), bwd_map), None, None


class RaggedPlan:
    """Everything the text tower derives from an attention mask: idx [Np] (flat positions of the real tokens, -1 padded to a multiple of the
    GEMM tile height: every row tile is full and the weight-gradient contraction length stays a multiple of 64), inv [B*L] (packed row of
    every position, -1 at padded positions), nv (number of real tokens), cu (int32 [B+1] first packed row of every sequence, or None when
    a mask has a hole - a real token behind a padded one - AND that is known: see ragged_plan)."""
    __slots__ = ("idx", "inv", "nv", "cu")

    def __init__(self, idx, inv, nv, cu):
        self.idx, self.inv, self.nv, self.cu = idx, inv, nv, cu


_RAGGED_KERNEL = os.environ.get("SIMSEG_AMD_RAGGED_KERNEL", "1") != "0"      # (A/B switch: 0 = the round-3 torch index ops with their two host reads)
_LENGTH_CHECKS = []       # (event, pinned info copy, what) of plans sized from host-side caption lengths: verified one step late, without a sync


_PINNED_INFO = []         # recycled pinned int32[4] buffers (pinning memory is a driver call: not once per step)


def _poll_length_checks(block=False):
    while _LENGTH_CHECKS and (block or len(_LENGTH_CHECKS) > 8 or _LENGTH_CHECKS[0][0].query()):
        ev, info, what = _LENGTH_CHECKS.pop(0)
        ev.synchronize()
        _PINNED_INFO.append(info)
        if int(info[2]):
            _LENGTH_CHECKS.clear()
            raise RuntimeError(f"caption_lengths do not describe the attention_mask of an earlier batch ({what}: the mask has {int(info[0])} real "
                               "tokens): the text tower of that step ran on the wrong rows")


def _plan_torch(mask, multiple):
    """The maps with torch index ops (CPU tensors in the host-logic tests, batches of more than 8192 sequences): two host reads."""
    flat = mask.reshape(-1) != 0
    idx = flat.nonzero().flatten().to(torch.int32)
    nv = idx.numel()
    inv = torch.full((flat.numel(),), -1, device=mask.device, dtype=torch.int32)
    inv[idx.long()] = torch.arange(nv, device=mask.device, dtype=torch.int32)
    pad = (-nv) % multiple
    if pad and nv + pad < flat.numel():
        idx = torch.cat([idx, torch.full((pad,), -1, device=mask.device, dtype=torch.int32)])
    real = mask != 0
    lens = real.sum(1)
    holes = (real & (torch.arange(mask.shape[1], device=mask.device)[None] >= lens[:, None])).any()
    cu = torch.zeros(mask.shape[0] + 1, device=mask.device, dtype=torch.int32)
    cu[1:] = torch.cumsum(lens, 0).to(torch.int32)
    return RaggedPlan(idx, inv, nv, None if bool(holes) else cu)


def ragged_plan(mask, multiple=256, lengths=None):
    """RaggedPlan of a [B, L] 0/1 mask, cached on the mask tensor (per version).  On the GPU ONE kernel builds all of it
    (ops.ragged_maps).  `lengths`: the captions' token counts as HOST numbers (a sequence / CPU tensor of B ints - a loader has them
    before the host->device copy): the packed row count is then known without reading anything back, and the step has no host
    synchronisation left in front of the loss.  The kernel still builds the maps from the device mask; its own count is compared with
    sum(lengths) on the device and checked one step late (RuntimeError then).  Without lengths the count (and the hole flag) are read
    back: one host read per mask tensor."""
    cached = getattr(mask, "_simseg_ragged", None)
    if cached is not None and cached[0] == mask._version and cached[1] == multiple:
        return cached[2]
    B, L = mask.shape
    if not (mask.is_cuda and _RAGGED_KERNEL and B <= 8192 and mask.dtype == torch.int64 and mask.is_contiguous()):
        plan = _plan_torch(mask, multiple)
    else:
        _poll_length_checks()
        if lengths is not None:
            nv = int(sum(int(v) for v in (lengths.tolist() if hasattr(lengths, "tolist") else lengths)))
            if len(lengths) != B or nv < 0 or nv > B * L:
                raise ValueError(f"caption_lengths: {len(lengths)} entries summing to {nv} for a [{B}, {L}] mask")
            np_ = nv + (-nv) % multiple
            cap = np_ if np_ < B * L else nv
            idx, inv, cu, info = ops.ragged_maps(mask, multiple, max(cap, 1), expect=nv)
            host = _PINNED_INFO.pop() if _PINNED_INFO else torch.empty(4, dtype=torch.int32).pin_memory()
            host.copy_(info, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            _LENGTH_CHECKS.append((ev, host, f"{B} captions, lengths summing to {nv}"))
            plan = RaggedPlan(idx[:cap], inv, nv, cu)      # (a hole does not matter to the packed attention: a sequence's real tokens are
                                                            #  contiguous rows either way; only the dropout hash would index them differently)
        else:
            idx, inv, cu, info = ops.ragged_maps(mask, multiple, B * L)
            nv, holes, _, _ = info.tolist()                 # the one host read
            np_ = nv + (-nv) % multiple                     # (no tile padding when the padded rows would not be fewer than the dense ones)
            plan = RaggedPlan(idx[:np_ if np_ < B * L else nv], inv, nv, None if holes else cu)
    try:
        mask._simseg_ragged = (mask._version, multiple, plan)
    except Exception:       # noqa: BLE001  (a tensor subclass without attribute storage)
        pass
    return plan


def ragged_maps(mask, multiple=256):
    """(idx, inv, number of real tokens) - see RaggedPlan."""
    p = ragged_plan(mask, multiple)
    return p.idx, p.inv, p.nv


def ragged_rows(mask):
    """int32 [B+1] first packed row of every sequence (the packed order keeps a sequence's real tokens together), or None when some mask
    has a hole.  The attention kernels then take the packed rows directly (simseg_attention_fwd_rows): no round trip through the dense
    [B, L] layout.  For prefix masks - every caption the reference's tokenizer produces - a token's index inside its packed sequence
    equals its position, so the dropout hash drops the same probabilities as the dense path."""
    return ragged_plan(mask).cu


_PACKED_ATTN = os.environ.get("SIMSEG_AMD_PACKED_ATTN", "1") != "0"      # (A/B switch: attention on the packed rows)


class BertLayerFn(_GradAwareFn):
    """HF BertLayer (post-LN, eps 1e-12): a = LN(x + drop(dense(attn(x)))); y = LN(a + drop(dense(gelu(dense(a)))))."""

    @staticmethod
    def forward(ctx, x, mask, heads, adt, drop_p, seed, qw, qb, kw, kb, vw, vb, ow, ob, law, lab, iw, ib, o2w, o2b, low, lob, idx=None, inv=None,
                cu=None, nv=0):
        # idx / inv given: x is [Nv, D], the real tokens of the ragged batch only (padded rows dropped).  Every GEMM, LayerNorm and
        # dropout then runs on Nv rows; only the attention kernels see the dense [B, L] layout (rows put back with zeros at the padded
        # positions: masked keys, and queries whose outputs are never read).
        packed = idx is not None
        B, L = mask.shape
        D = x.shape[-1]
        x = x.contiguous()
        save = _saving(ctx)
        # (bf16: the previous layer's output LayerNorm has written a bf16 copy of x next to the fp32 rows - no cast pass)
        sh16 = getattr(x, "_simseg_fwd16", None) if adt != F32 else None
        if sh16 is not None and sh16[1] == x._version and sh16[0].numel() == x.numel():
            xa = sh16[0].view(-1, D)
        else:
            xa = x.view(-1, D) if adt == F32 else ops.cast(x.view(-1, D), adt)
        wqkv = _wt_stacked((qw, kw, vw), adt)                              # HF keeps three matrices; one fused [3D, D] GEMM here
        bqkv = _as_one(qb.detach(), kb.detach(), vb.detach())
        if bqkv is None:
            bqkv = torch.cat([qb.detach(), kb.detach(), vb.detach()])
        ow_, iw_, o2w_ = _wt(ow, adt), _wt(iw, adt), _wt(o2w, adt)
        qkv = ops.gemm(xa, wqkv, bias=bqkv)
        rows = packed and cu is not None and adt != F32             # attention straight on the packed rows
        if rows:
            att, lse = ops.attention_fwd_rows(qkv, heads, cu, L, scale=64 ** -0.5, save_lse=save, drop_seed=seed, drop_p=drop_p, n_real=nv)
            attd = None
        else:
            if packed:
                qkv = ops.gather_rows(qkv, inv)                      # [B*L, 3D], zero rows at the padded positions
            att, lse = ops.attention_fwd(qkv.view(B, L, 3 * D), heads, mask, scale=64 ** -0.5, save_lse=save, drop_seed=seed, drop_p=drop_p,
                                          skip_padded_rows=packed and _SKIP_PAD, zero_skipped=False)      # (only real rows are gathered below)
            attd = att
            if packed:
                att = ops.gather_rows(att.view(-1, D), idx)          # [Nv, D]
        s1 = _fwd_gemm(att.view(-1, D), ow, adt, save, bias=ob.detach(), residual=x.view(-1, D), out_dtype=F32, drop_seed=seed + 1, drop_p=drop_p)
        a32, a16, mean_a, rstd_a = ops.layernorm_fwd(s1, law.detach(), lab.detach(), 1e-12, want_bf16_copy=(adt if adt != F32 else False), save_stats=save)
        aa = a32 if adt == F32 else a16
        blk = save and adt != F32 and ops.gemm_aux_blocked_ok(aa.shape[0], iw.shape[0], D)
        g8 = blk and _GELU8
        pre = torch.empty(x.shape[0] if packed else B * L, iw.shape[0], device=x.device, dtype=torch.uint8 if g8 else adt) if save else None
        act = _fwd_gemm(aa, iw, adt, save, bias=ib.detach(), act=((7 if g8 else 5) if blk else 3) if save else 1, aux_out=pre)       # pre holds GELU'(intermediate)
        s2 = _fwd_gemm(act, o2w, adt, save, bias=o2b.detach(), residual=a32, out_dtype=F32, drop_seed=seed + 2, drop_p=drop_p)
        y, y16, mean_o, rstd_o = ops.layernorm_fwd(s2, low.detach(), lob.detach(), 1e-12, want_bf16_copy=(adt if (adt != F32 and packed) else False), save_stats=save)
        if y16 is not None:
            y._simseg_fwd16 = (y16, y._version)                            # picked up by the next layer's forward (same tensor object)
        ctx.adt, ctx.heads, ctx.dims, ctx.drop, ctx.packed, ctx.rows, ctx.nv, ctx.blk = adt, heads, (B, L, D), (drop_p, seed), packed, rows, nv, (2 if g8 else 1) if blk else 0
        ctx.wparams = (o2w, iw, ow)                                       # (the parameter objects: _grad_target in the backward)
        if save:
            ctx.save_for_backward(xa, mask, qkv, att, lse, s1, mean_a, rstd_a, aa, pre, act, s2, mean_o, rstd_o, wqkv, ow_, iw_, o2w_,
                                  law.detach(), low.detach(), attd if (packed and not rows) else None, idx, inv, cu if rows else None)
        return y if packed else y.view(B, L, D)

    @staticmethod
    @_joins_wgrad
    def backward(ctx, dy):
        (xa, mask, qkv, att, lse, s1, mean_a, rstd_a, aa, pre, act, s2, mean_o, rstd_o, wqkv, ow_, iw_, o2w_, law, low, attd, idx, inv, cu) = ctx.saved_tensors
        B, L, D = ctx.dims
        packed = ctx.packed
        adt = ctx.adt
        p, seed = ctx.drop
        need = ctx.needs_input_grad
        dy = dy.contiguous().view(-1, D)
        I = pre.shape[1]
        tg = [_grad_target(w) if nd else None for w, nd in zip(ctx.wparams, (need[18], need[16], need[12]))]
        (dlow, dlob, do2b, dib, do2w_z, diw_z, dlaw, dlab, dob, dow_z, dwqkv_z, dbqkv_z) = _zeros_or(
            dy.device, (None, None, None, None, tg[0], tg[1], None, None, None, tg[2], None, None),
            (D,), (D,), (D,), (I,), (D, I), (I, D), (D,), (D,), (D,), (D, D), (3 * D, D), (3 * D,))
        ds2_32, d2 = _ln_bwd(adt, s2, mean_o, rstd_o, low, dlow, dlob, None if adt != F32 else dy, dy32=dy if adt != F32 else None,
                             dxsum=do2b, drop=(p, seed + 2))
        dpre = _dgrad(d2, o2w_, act=(4, 6, 8)[ctx.blk], aux=pre, colsum=dib)
        do2w = _wgrad(d2, act, do2w_z) if need[18] else None
        # gradient reaching LN_a's output: through the intermediate dense (da) + the residual branch (ds2_32); bf16 mode adds them
        # inside the LayerNorm kernel, exact mode in the GEMM's residual epilogue
        da = _dgrad(dpre, iw_) if adt != F32 else _dgrad(dpre, iw_, residual=ds2_32)
        diw = _wgrad(dpre, aa, diw_z) if need[16] else None
        ds1_32, d1 = _ln_bwd(adt, s1, mean_a, rstd_a, law, dlaw, dlab, da, dy32=ds2_32 if adt != F32 else None, dxsum=dob,
                             drop=(p, seed + 1))
        datt = _dgrad(d1, ow_)
        dow = _wgrad(d1, att.view(-1, D), dow_z) if need[12] else None
        dbqkv = _zeros_like_bias(dbqkv_z, 3 * D, qkv.device) if (need[7] or need[9] or need[11]) else None
        if ctx.rows:                                                 # attention backward straight on the packed rows
            dqkv = ops.attention_bwd_rows(qkv, att, datt.contiguous(), lse, ctx.heads, cu, L, scale=64 ** -0.5, drop_seed=seed, drop_p=p,
                                          colsum=dbqkv if _FUSED_QKV_BIAS else None, n_real=ctx.nv)
        else:
            if packed:
                datt = ops.gather_rows(datt, inv)                    # back to [B*L, D] (zero rows at the padded positions)
            # (packed: rows of padded tokens carry no gradient - their dout is zero, their keys are masked - so the sums over the dense rows
            # the kernel works on equal the sums over the packed rows)
            dqkv = ops.attention_bwd(qkv.view(B, L, 3 * D), (attd if packed else att).view(B, L, D), datt.view(B, L, D), lse, ctx.heads, mask,
                                     scale=64 ** -0.5, drop_seed=seed, drop_p=p, skip_padded_rows=packed and _SKIP_PAD,
                                     colsum=dbqkv if _FUSED_QKV_BIAS else None, zero_skipped=False).view(-1, 3 * D)
            if packed:
                dqkv = ops.gather_rows(dqkv, idx)                    # [Nv, 3D]
        if dbqkv is not None and not _FUSED_QKV_BIAS:
            ops.colsum_accum(dqkv, dbqkv)
        dx = _dgrad(dqkv, wqkv, residual=ds1_32, out_dtype=F32)
        dwqkv = _wgrad(dqkv, xa, dwqkv_z) if (need[6] or need[8] or need[10]) else None
        dws = [dwqkv[i * D:(i + 1) * D] if dwqkv is not None else None for i in range(3)]
        dbs = [dbqkv[i * D:(i + 1) * D] if dbqkv is not None else None for i in range(3)]
        return (dx if packed else dx.view(B, L, D), None, None, None, None, None, dws[0], dbs[0], dws[1], dbs[1], dws[2], dbs[2], dow, dob,
                dlaw, dlab, diw, dib, do2w, do2b, dlow, dlob, None, None, None, None)


# Ragged caption batches.  HF's BertModel computes every padded token (huggingface_builder.py:16-17); nothing downstream of the CLIP
# pipeline reads those rows (the key-padding mask hides them from the real tokens, the masked top-1 pooling of forward_text_project
# drops them, their gradients are exactly zero).  Inside `packed_text()` - which CLIPModel.forward enters around its text branch - the
# tower therefore runs its GEMMs, LayerNorms and dropout on the real tokens only and returns zeros at the padded positions; loss,
# accuracies and every parameter gradient are those of the dense computation.  The plain `forward_text_feature` API stays dense, so its
# [B, L, 768] output equals the reference's at every position.  SIMSEG_AMD_PACKED_TEXT=0 switches the packing off.
_PACK_TEXT = [False, None]       # [pack the ragged batch?, host-side caption lengths of the batch being encoded (or None)]
_SKIP_PAD = os.environ.get("SIMSEG_AMD_SKIP_PAD_ROWS", "1") != "0"      # (A/B switch: attention kernels on the effective lengths)


@contextlib.contextmanager
def packed_text(lengths=None):
    """lengths: the batch's caption token counts as host numbers (batch["caption_lengths"], optional) - see ragged_plan."""
    prev = list(_PACK_TEXT)
    _PACK_TEXT[0] = os.environ.get("SIMSEG_AMD_PACKED_TEXT", "1") != "0"
    _PACK_TEXT[1] = lengths
    try:
        yield
    finally:
        _PACK_TEXT[:] = prev


def bert_forward(m, input_ids, attention_mask, adt, training=False, seed=0):
    """m: module tree with HF BertModel parameter names (simseg_amd/nn.py Bert). Returns last_hidden_state [B,L,D] fp32."""
    p_h = m.hidden_dropout_prob if training else 0.0
    p_a = m.attention_probs_dropout_prob if training else 0.0
    if p_h != p_a:
        raise NotImplementedError("hidden and attention dropout probabilities are expected to be equal (HF default 0.1)")
    e = m.embeddings
    mask = attention_mask.contiguous().long()
    x = BertEmbedFn.apply(input_ids, mask, e.word_embeddings.weight, e.position_embeddings.weight, e.token_type_embeddings.weight,
                          e.LayerNorm.weight, e.LayerNorm.bias, p_h, seed)
    B, L, D = x.shape
    idx = inv = cu = None
    nv = 0
    if _PACK_TEXT[0] and x.is_cuda:
        lengths = _PACK_TEXT[1]
        plan = ragged_plan(mask, lengths=lengths if (lengths is not None and len(lengths) == B) else None)
        idx, inv, nv = plan.idx, plan.inv, plan.nv
        if nv == 0 or idx.numel() >= B * L:
            idx = inv = None                                        # nothing to drop
        else:
            x = RowMapFn.apply(x.view(-1, D), idx, inv)             # [Nv, D]
            if _PACKED_ATTN and adt != F32 and L <= 256:
                cu = plan.cu                                        # None: a mask with a hole - attention through the dense layout
    for i, lyr in enumerate(m.encoder.layer):
        a, s = lyr.attention, lyr.attention.self
        x = BertLayerFn.apply(x, mask, m.num_heads, adt, p_h, seed + 16 * (i + 1),
                              s.query.weight, s.query.bias, s.key.weight, s.key.bias, s.value.weight, s.value.bias,
                              a.output.dense.weight, a.output.dense.bias, a.output.LayerNorm.weight, a.output.LayerNorm.bias,
                              lyr.intermediate.dense.weight, lyr.intermediate.dense.bias,
                              lyr.output.dense.weight, lyr.output.dense.bias, lyr.output.LayerNorm.weight, lyr.output.LayerNorm.bias,
                              idx, inv, cu, nv)
    if idx is not None:
        x = RowMapFn.apply(x, inv, idx).view(B, L, D)               # zeros at the padded positions
    return x


# ------------------------------------------------------------------------------------------------------------------
# heads
# ------------------------------------------------------------------------------------------------------------------
_SKIP_MASKS = {}


def _skip_mask(B, N, skip, device):
    key = (B, N, skip, str(device))
    m = _SKIP_MASKS.get(key)
    if m is None:
        if len(_SKIP_MASKS) > 16:
            _SKIP_MASKS.clear()
        m = torch.ones(B, N, device=device, dtype=torch.int64)
        m[:, :skip] = 0
        _SKIP_MASKS[key] = m
    return m


class ProjectPoolFn(_GradAwareFn):
    """SimpleProjection -> TopKPooling(LoDA) -> L2norm  (pipelines/clip.py:87-93, 111-120) as one node:
    the [B,N,512] token projection lives only inside this function."""

    @staticmethod
    def forward(ctx, feats, w, k, mask, adt, skip=0):
        # skip = s: the first s tokens of every sequence take no part in the pooling (the image tower's [cls] token: the reference pools
        # feats[:, 1:], clip.py:65-84).  Handing the tower's whole output over and masking the token here saves the slice's copy of the
        # features in the forward, the zero-filled scatter of its gradient in the backward and - with the 16-bit copy the tower's last
        # LayerNorm wrote (feats._simseg_fwd16) - the cast: four passes over [B, N, D] per step.  Same pooled values, same gradients.
        B, N, D = feats.shape
        if skip:
            if mask is not None:
                raise ValueError("ProjectPoolFn: skip and a token mask together are not supported")
            mask = _skip_mask(B, N, int(skip), feats.device)
        sh16 = getattr(feats, "_simseg_fwd16", None) if adt != F32 else None
        if sh16 is not None and sh16[1] == feats._version and sh16[0].shape == feats.shape and sh16[0].dtype == adt:
            fa = sh16[0].view(-1, D)
        else:
            f2 = feats.contiguous().view(-1, D)
            fa = f2 if adt == F32 else ops.cast(f2, adt)
        ctx.adt, ctx.k, ctx.dims = adt, k, (B, N, D)
        if not _saving(ctx):
            tok = _fwd_gemm(fa, w, adt, False).view(B, N, w.shape[0])
            return ops.topk_pool_l2norm_fwd(tok, k, mask)[0]
        wa = _wt(w, adt)
        tok = ops.gemm(fa, wa).view(B, N, w.shape[0])
        emb, idx, norm = ops.topk_pool_l2norm_fwd(tok, k, mask)
        ctx.save_for_backward(fa, wa, emb, idx, norm)
        return emb

    @staticmethod
    @_joins_wgrad
    def backward(ctx, demb):
        fa, wa, emb, idx, norm = ctx.saved_tensors
        B, N, D = ctx.dims
        dtok = ops.topk_pool_l2norm_bwd(demb.contiguous(), emb, norm, idx, N, ctx.adt).view(B * N, -1)
        dfe = _dgrad(dtok, wa, out_dtype=F32).view(B, N, D) if ctx.needs_input_grad[0] else None
        dw = _wgrad(dtok, fa) if ctx.needs_input_grad[1] else None
        return dfe, dw, None, None, None, None
