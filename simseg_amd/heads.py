"""Loss / similarity heads on the HIP kernels: differentiable embedding all-gather, global InfoNCE, the dense
patch x class-text similarity map, and the retrieval recall computation."""
import weakref

import torch
import torch.distributed as dist
from torch.autograd import Function

from . import ops

F32 = torch.float32


# Benchmarks only (bench.py's "communication-free" leg at N > 1 ranks): every embedding exchange is replaced by a LOCAL stand-in of the
# same shapes - the gathered matrix is this rank's rows repeated `world` times, the reduce-scatter keeps this rank's slice - so the step
# computes everything it computes with the exchange (the [Bl, W * Bl] similarity blocks included) and moves no byte between ranks.  The
# numbers are not the global loss's; what the leg is for is step(N) - step(N, stand-ins) = the communication the step could not hide.
LOCAL_STANDIN = False


def _all_gather(out, t, group, async_op=False):
    if LOCAL_STANDIN:
        out.view((-1,) + tuple(t.shape)).copy_(t.unsqueeze(0).expand((out.shape[0] // t.shape[0],) + tuple(t.shape)))
        return None
    return dist.all_gather_into_tensor(out, t, group=group, async_op=async_op)


def _reduce_scatter(own, grad, group):
    if LOCAL_STANDIN:
        own.copy_(grad.view((-1,) + tuple(own.shape))[dist.get_rank(group)])
        return
    dist.reduce_scatter_tensor(own, grad, op=dist.ReduceOp.SUM, group=group)


def _one(world):
    """The one-rank shortcut applies (see parallel.force_collectives: SIMSEG_FORCE_COLLECTIVES=1 keeps the collectives at world size 1)."""
    from .parallel import force_collectives
    return world == 1 and not force_collectives()


# Embedding all-gathers started ahead of their use (prefetch_gather): id(tensor) -> (weakref, input, output, work).  The exchange
# of one tower's [Bl, P] embeddings then runs on the collective's own stream underneath the OTHER tower's kernels.
_PREFETCH = {}


def _gathered_shape(t, world):
    return (world * t.shape[0],) + tuple(t.shape[1:])


def prefetch_gather(tensor, group=None):
    """Start the all-gather of `tensor`'s rows over `group` now (asynchronously, in rank order); the next GatherLayer /
    all_gather_rows call on the same tensor object picks the result up instead of issuing its own collective."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    world = dist.get_world_size(group)
    if _one(world) or LOCAL_STANDIN:
        return
    t = tensor.detach().contiguous()
    out = torch.empty(_gathered_shape(t, world), device=t.device, dtype=t.dtype)
    work = dist.all_gather_into_tensor(out, t, group=group, async_op=True)
    _PREFETCH[id(tensor)] = (weakref.ref(tensor), t, out, work, group)


def discard_prefetched(log=None):
    """Drop every prefetched gather nobody picked up (the loss was not reached - an exception, or a custom loss that normalises / casts
    the embeddings before gathering them): the collective is waited for so that its buffers can be recycled, and the entry is removed.
    CLIPModel.forward calls this first, so a missed hand-over costs one redundant collective once instead of leaking a buffer and a Work
    handle every step.  Returns the number of discarded entries."""
    n = 0
    for key in list(_PREFETCH):
        ent = _PREFETCH.pop(key, None)
        if ent is None:
            continue
        try:
            ent[3].wait()
        except Exception:       # noqa: BLE001  (a failed collective of an abandoned step)
            pass
        n += 1
    if n and log is not None:
        log(f"simseg_amd.heads: {n} prefetched embedding gather(s) were never consumed and have been discarded")
    return n


def _take_prefetched(tensor, group):
    ent = _PREFETCH.pop(id(tensor), None)
    if ent is None or ent[0]() is not tensor or ent[4] is not group:
        return None
    ent[3].wait()                     # the compute stream waits for the collective's stream; the host does not block (RCCL)
    if ent[2].is_cuda:                # allocated under the producing tower's stream, consumed on the loss's stream
        ent[2].record_stream(torch.cuda.current_stream())
    return ent[2]


class GatherLayer(Function):
    """Differentiable all-gather of [Bl, ...] rows in rank order (simseg/utils/dist.py:323-354).
    Forward: all-gather into one [W*Bl, ...] buffer.  Backward: the reference all-reduces the whole gathered gradient and
    slices its own rows (:347-354); a reduce-scatter delivers exactly those rows with 1/W of the traffic.  One code path for
    every backend (RCCL on GPUs; gloo in the CPU tests implements both collectives too)."""

    @staticmethod
    def forward(ctx, tensor, group, rank):
        ctx.group, ctx.rank, ctx.bl = group, rank, tensor.shape[0]
        world = dist.get_world_size(group)
        out = _take_prefetched(tensor, group)
        if out is not None:
            return out
        tensor = tensor.contiguous()
        out = torch.empty(_gathered_shape(tensor, world), device=tensor.device, dtype=tensor.dtype)
        if _one(world):
            out.copy_(tensor)
        else:
            _all_gather(out, tensor, group)
        return out

    @staticmethod
    def backward(ctx, grad):
        world = dist.get_world_size(ctx.group)
        grad = grad.contiguous()
        if _one(world):
            return grad.clone(), None, None
        own = torch.empty((ctx.bl,) + tuple(grad.shape[1:]), device=grad.device, dtype=grad.dtype)
        _reduce_scatter(own, grad, ctx.group)
        return own, None, None


def all_gather_rows(tensor, group):
    """Non-differentiable variant (simseg/utils/dist.py:65-74, gather_backward=False)."""
    world = dist.get_world_size(group)
    out = _take_prefetched(tensor, group)
    if out is not None:
        return out
    tensor = tensor.detach().contiguous()
    if _one(world):
        return tensor
    out = torch.empty(_gathered_shape(tensor, world), device=tensor.device, dtype=tensor.dtype)
    _all_gather(out, tensor, group)
    return out


class NCEFn(Function):
    """One direction of the global InfoNCE (mml_loss.py:51-103): feat1 [Bl,P] against feat2_global [Bg,P].
    Returns (loss, top-1 acc) as 0-dim tensors.  The [Bl,Bg] similarity block is produced by the fp32 MFMA GEMM and
    overwritten in place by its own gradient, so forward+backward touch it twice."""

    @staticmethod
    def forward(ctx, feat1, feat2g, temperature, ignore, ignore_g, rank, smoothing):
        f1 = feat1.contiguous().float()
        f2 = feat2g.contiguous().float()
        if ignore_g is not None:
            f2 = ops.scale_rows(f2, ignore_g.contiguous().float(), one_minus=True)       # :70-71
        sims = ops.gemm(f1, f2)
        need = any(ctx.needs_input_grad[:3])
        out3 = ops.nce_rows(sims, temperature.detach().reshape(1).float(), rank * f1.shape[0],
                            None if ignore is None else ignore.contiguous().float(), smoothing, write_grad=need)
        ctx.save_for_backward(sims if need else None, f1, f2, out3, ignore_g)
        loss, acc = out3[0].clone(), out3[1].clone()
        ctx.mark_non_differentiable(acc)
        return loss, acc

    @staticmethod
    def backward(ctx, gloss, _gacc):
        ds, f1, f2, out3, ignore_g = ctx.saved_tensors
        g = gloss.contiguous().float().reshape(1)
        df1 = df2 = dt = None
        if ctx.needs_input_grad[0]:
            df1 = ops.scale_by_scalar(ops.gemm(ds, ops.transpose_f32(f2)), g)
        if ctx.needs_input_grad[1]:
            df2 = ops.gemm(ops.transpose_f32(ds), ops.transpose_f32(f1))
            if ignore_g is not None:
                df2 = ops.scale_rows(df2, ignore_g.contiguous().float(), one_minus=True)
            df2 = ops.scale_by_scalar(df2, g)
        if ctx.needs_input_grad[2]:
            dt = ops.scale_by_scalar(out3[2:3], g).reshape(())
        return df1, df2, dt, None, None, None, None


class ClipLossFn(Function):
    """The whole loss head of CLIPModel.forward_loss (pipelines/clip.py:129-140) as ONE autograd node: 0.5 * (NCE(image, text) +
    NCE(text, image)) with the global-batch exchange inside (mml_loss.py:56-77: all-gather of the other modality's embeddings forward,
    reduce-scatter of their gradients backward - GatherLayer's arithmetic).  Same kernels as NCEFn (exact fp32 MFMA similarity blocks,
    in-place cross-entropy rows), fewer launches: forward = two GEMMs + one rows launch over both blocks + one finalize; backward = one
    launch that scales both gradient blocks by the upstream gradient and makes every transposed GEMM operand, then four GEMMs (the
    second of each pair accumulating into the first's result when there is one rank).  The round-3 head took ~14 + ~22 launches with
    ~160 us holes between 4-11 us kernels.  Returns (loss, i2t acc, t2i acc)."""

    @staticmethod
    def forward(ctx, img, txt, temperature, group, rank, smoothing, gather_backward):
        img32, txt32 = img.contiguous().float(), txt.contiguous().float()
        world = dist.get_world_size(group) if group is not None else 1
        multi = group is not None and not _one(world)          # the exchange runs: several ranks, or a one-rank group with SIMSEG_FORCE_COLLECTIVES=1
        if multi:
            img_g = _take_prefetched(img, group)
            txt_g = _take_prefetched(txt, group)
            if img_g is None:
                img_g = torch.empty(_gathered_shape(img32, world), device=img.device, dtype=F32)
                _all_gather(img_g, img32, group)
            if txt_g is None:
                txt_g = torch.empty(_gathered_shape(txt32, world), device=txt.device, dtype=F32)
                _all_gather(txt_g, txt32, group)
            img_g, txt_g = img_g.float(), txt_g.float()
        else:
            img_g, txt_g = img32, txt32
        Bl, Bg = img32.shape[0], img_g.shape[0]
        sims = torch.empty(2, Bl, Bg, device=img.device, dtype=F32)
        ops.gemm(img32, txt_g, out=sims[0])
        ops.gemm(txt32, img_g, out=sims[1])
        need = any(ctx.needs_input_grad[:3])
        out4 = ops.nce_pair(sims, temperature.detach().reshape(1).float(), rank * Bl, smoothing, write_grad=need)
        ctx.group, ctx.world, ctx.rank, ctx.gb, ctx.multi = group, world, rank, gather_backward, multi
        ctx.save_for_backward(sims if need else None, img32, txt32, img_g if multi else None, txt_g if multi else None, out4)
        loss, a1, a2 = out4[0].clone(), out4[1].clone(), out4[2].clone()
        ctx.mark_non_differentiable(a1, a2)
        return loss, a1, a2

    @staticmethod
    def backward(ctx, gloss, _g1, _g2):
        ds, img32, txt32, img_g, txt_g, out4 = ctx.saved_tensors
        one = not ctx.multi                                     # no exchange: the "gathered" embeddings are the local ones
        g = gloss.contiguous().float().reshape(1)
        need_img, need_txt, need_t = ctx.needs_input_grad[:3]
        x0 = out4[3:4] if need_t else None
        # ONE launch: both gradient blocks times 0.5 * g (in place and transposed) + the transposed embeddings; dT * g on the way
        if one:
            (dsi_t, dst_t, txt_t, img_t), dt = ops.transpose_multi([ds[0], ds[1], txt32, img32], [1, 1, 0, 0], scalar=g, alpha=0.5, x0=x0)
            txtg_t, imgg_t = txt_t, img_t
        else:
            (dsi_t, dst_t, txtg_t, imgg_t, txt_t, img_t), dt = ops.transpose_multi([ds[0], ds[1], txt_g, img_g, txt32, img32], [1, 1, 0, 0, 0, 0],
                                                                                    scalar=g, alpha=0.5, x0=x0)
        dimg = dtxt = None
        # gradients flow through the gathered role with gather_backward, and with NO process group (NCE._gather then returns the local
        # tensor itself).  A real group of one rank with gather_backward=False detaches like all_gather_rows / the reference's
        # all_gather_group (utils/dist.py:65-74) - same gradients as the unfused NCE.forward path.
        through_gather = ctx.gb or ctx.group is None
        if need_img:
            dimg = ops.gemm(ds[0], txtg_t)                                   # dS_i . txt_g          [Bl, P]
            if through_gather:
                if one:
                    ops.gemm(dst_t, txt_t, out=dimg, accumulate=True)        # + dS_t^T . txt
                else:
                    dimg_g = ops.gemm(dst_t, txt_t)                          # [Bg, P]: every rank's rows; this rank keeps the sum of its own
                    own = torch.empty_like(dimg)
                    _reduce_scatter(own, dimg_g, ctx.group)
                    dimg = dimg + own
        if need_txt:
            dtxt = ops.gemm(ds[1], imgg_t)                                   # dS_t . img_g
            if through_gather:
                if one:
                    ops.gemm(dsi_t, img_t, out=dtxt, accumulate=True)        # + dS_i^T . img
                else:
                    dtxt_g = ops.gemm(dsi_t, img_t)
                    own = torch.empty_like(dtxt)
                    _reduce_scatter(own, dtxt_g, ctx.group)
                    dtxt = dtxt + own
        return dimg, dtxt, (dt.reshape(()) if need_t else None), None, None, None, None


def patch_text_similarity(patch_proj, text_feat, eps=1e-12, compute_dtype=F32):
    """sim[b,n,c] = <normalize(patch_proj[b,n,:]), text_feat[c,:]>  -- the dense zero-shot segmentation map
    (tools/seg_evaluation.py:112 F.normalize + :136 per-class GEMV, here for every class at once).
    The row normalisation is fused into the GEMM epilogue as a row scale."""
    B, N, P = patch_proj.shape
    x = patch_proj.contiguous().view(B * N, P)
    t = text_feat.contiguous()
    want = compute_dtype if compute_dtype in ops.HALF_TYPES else F32
    x = x if x.dtype == want else (ops.cast(x.float(), want) if want != F32 else x.float())
    t = t if t.dtype == want else (ops.cast(t.float(), want) if want != F32 else t.float())
    epc = 8 if want != F32 else 4
    if t.shape[0] <= 256 and P % epc == 0:
        return ops.patch_text_sim(x, t, eps).view(B, N, t.shape[0])      # fused: x is read once
    rn = ops.row_rnorm(x, eps)                                           # > 256 classes: row scale in the GEMM epilogue
    return ops.gemm(x, t, rowscale=rn, out_dtype=F32).view(B, N, t.shape[0])


def _similarity_matrix(left, right):
    """left @ right^T in fp32 (the reference's torch.matmul of the gathered embeddings, tasks/clip/hooks/utils.py:36).  Large problems - the
    5000 x 25000 x 512 matrix of the MSCOCO-5k evaluation - go through the split-bf16 form of the fp32 product (towers._fwd_gemm: six
    bf16 piece products in one MFMA launch, fp32-grade accuracy) instead of the fp32 MFMA kernel."""
    from . import towers
    a, b = left.contiguous().float(), right.contiguous().float()
    if towers._split_ok(a.shape[0], b.shape[0], a.shape[1]):
        towers.SPLIT_CALLS[0] += 1
        return ops.gemm(ops.split_bf16x3(a), ops.split_bf16x3(b, b_pattern=True), out_dtype=F32)
    return ops.gemm(a, b)


def retrieval_recalls(left, left_gid, right, right_gid, bounds=(1, 5, 10)):
    """R@k of `left` rows retrieving `right` rows sharing their group id (hooks/utils.py:59-75).  One host sync
    (4 counters), like the reference's .item()."""
    sim = _similarity_matrix(left, right)
    has, rank = ops.retrieval_rank(sim, left_gid.contiguous().long(), right_gid.contiguous().long())
    c = ops.recall_counts(has, rank, bounds).cpu()
    if int(c[0]) == 0:
        raise AssertionError("no left row has a matching right row")
    return {f"R@{b}": float(c[i + 1]) / float(c[0]) for i, b in enumerate(bounds)}


def retrieval_recalls_both(left, left_gid, right, right_gid, bounds=(1, 5, 10)):
    """Both directions of the retrieval evaluation (tools/retrieval_evaluation.py:33-40 calls the metric twice with swapped arguments)
    from ONE similarity matrix: rows rank their columns, columns rank their rows.  Returns (left->right, right->left)."""
    lg, rg = left_gid.contiguous().long(), right_gid.contiguous().long()
    sim = _similarity_matrix(left, right)
    has, rank = ops.retrieval_rank(sim, lg, rg)
    hasc, rankc = ops.retrieval_rank_cols(sim, lg, rg)
    c = torch.stack([ops.recall_counts(has, rank, bounds), ops.recall_counts(hasc, rankc, bounds)]).cpu()
    out = []
    for d in range(2):
        if int(c[d, 0]) == 0:
            raise AssertionError("no row has a match in the other set")
        out.append({f"R@{b}": float(c[d, i + 1]) / float(c[d, 0]) for i, b in enumerate(bounds)})
    return out[0], out[1]


@torch.no_grad()
def class_text_embeddings(model, input_ids, attention_mask, chunk=2048):
    """Zero-shot classifier weights (tools/seg_evaluation.py:57-75) as batched calls: input_ids / attention_mask are
    [C, P, L] (C classes x P prompt templates); the text tower runs over C*P captions in chunks and one kernel reduces the
    prompt ensemble (mean over P, re-normalise).  Returns [C, proj_dim] unit-norm rows."""
    C, Pn, L = input_ids.shape
    ids, mask = input_ids.reshape(C * Pn, L), attention_mask.reshape(C * Pn, L)
    embs = []
    from .towers import packed_text
    for s in range(0, C * Pn, chunk):
        with packed_text():          # only the masked pooling reads the tower's output: the prompts' padded token rows are not computed
            feats = model.forward_text_feature(ids[s:s + chunk], mask[s:s + chunk])
            embs.append(model.forward_text_project(feats, mask[s:s + chunk]))
    emb = torch.cat(embs).float().view(C, Pn, -1)
    return ops.segment_mean_l2norm(emb.contiguous())
