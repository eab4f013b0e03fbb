"""Per-image body of the reference's `evaluate_benchmark` (tools/seg_evaluation.py:99-170) as batched device work.

    feats  = model.forward_image_feature(image)              # [B, n*n, D]
    pooled = model.forward_image_project(feats)              # [B, 512]
    sim    = patch_text_similarity(model.image_projection(feats), text)      # [B, n*n, C]   (K14)
    out    = segment(sim, pooled @ text.T, labels, n, top_cls_num)

Between the normalised map and the morphology the reference runs a fully connected CRF on the CPU for every visited candidate
(pydensecrf, :31-54 / :153).  Here it is `ops.dense_crf` - mean-field inference on two permutohedral lattices built on the device -
when the caller hands over the de-normalised network inputs (`images_u8`, what the tool builds at :104); without them the binary
map is the CRF's unary decision (prob > 0.5), and a caller-supplied `refine` hook can replace either.  Everything after it (7x7
dilate + erode, nearest resize, score-weighted argmax, IoU histograms) is the reference's arithmetic again.
"""
import os

import torch

from . import ops


CRF_PARAMS = dict(sxy_g=3.0, compat_g=3.0, sxy_b=40.0, srgb=13.0, compat_b=10.0, iters=3)        # tools/seg_evaluation.py:48-51


def crf_masks(prob, cand_idx, images_u8, chunk=None, scale=1, static=None, **params):
    """prob [B,K,h,w] fp32 normalised candidate maps (at 1/scale of the image resolution: the x16 nearest upsampling of the tool,
    tools/seg_evaluation.py:141-143, is applied here per chunk and only to visited slots), cand_idx [B,K] (-1 = slot not visited),
    images_u8 [B,H,W,3] -> uint8 masks [B,K,H,W] (0/255; unvisited slots zero).  One host read of the candidate table per batch (the
    reference loops on the host per image and candidate); images with a visited candidate go through the device CRF `chunk` at a time
    (an image's candidates share its lattices; images are grouped by their number of visited candidates, so chunks carry few idle maps).
    Per chunk: one gather of the visited low-resolution maps, one upsampling, one CRF call, one scatter of the masks."""
    B, K, h, w = prob.shape
    H, W = h * scale, w * scale
    dev = prob.device
    if static is None:
        static = os.environ.get("SIMSEG_CRF_STATIC", "0") != "0"
    if static:
        # Sync-free form (opt-in): no host read and no data-dependent shapes - every image goes through the CRF in ONE call with all K
        # candidate slots as channels; slots the reference never visits carry the all-zero map seg_masks leaves there and their result is
        # dropped.  K = 5 channels instead of the ~2 visited maps per image: measured 55.6 vs 28.4 ms per 63-window batch of 512^2
        # (round 3), so the chunked form below stays the default.  (Replaying the CRF inside a hipGraph is NOT supported: a first
        # attempt hung in replay - the hash build's probe loops never terminate if a table memset is not replayed - and was not pursued.)
        up = prob if scale == 1 else prob.repeat_interleave(scale, 2).repeat_interleave(scale, 3)
        m, _ = ops.dense_crf(images_u8.contiguous(), up.contiguous(), **dict(CRF_PARAMS, **params))
        return m * (cand_idx >= 0).to(torch.uint8)[:, :, None, None]
    if chunk is None:
        chunk = int(os.environ.get("SIMSEG_CRF_CHUNK", "16"))
    masks = torch.zeros(B, K, H, W, device=dev, dtype=torch.uint8)
    visited = (cand_idx >= 0).cpu()
    kw = dict(CRF_PARAMS, **params)
    todo = [(b, visited[b].nonzero().flatten().tolist()) for b in range(B)]
    todo = sorted(((b, ks) for b, ks in todo if ks), key=lambda t: len(t[1]))      # chunks of equal width: no idle zero maps
    for s in range(0, len(todo), chunk):
        part = todo[s:s + chunk]
        cmax = max(len(ks) for _, ks in part)
        slot = torch.full((len(part), cmax), -1, dtype=torch.int64)
        for j, (_, ks) in enumerate(part):
            slot[j, :len(ks)] = torch.tensor(ks)
        valid = (slot >= 0).to(dev)
        bsel = torch.tensor([b for b, _ in part], device=dev)
        ksel = slot.clamp(min=0).to(dev)                           # idle slots repeat a visited map (their result is dropped)
        lo = prob[bsel[:, None], ksel]                             # [n, cmax, h, w]
        up = lo if scale == 1 else lo.repeat_interleave(scale, 2).repeat_interleave(scale, 3)
        m, _ = ops.dense_crf(images_u8[bsel].contiguous(), up.contiguous(), **kw)
        masks[bsel[:, None].expand(-1, cmax)[valid], ksel[valid]] = m[valid]
    return masks


def segment(sim, scores, labels, num_patch, top_cls_num, num_classes=None, ncand=5, ignore_index=255, refine=None, hist=None,
            want_pred=True, closing=True, images_u8=None):
    """sim [B,n*n,C] fp32, scores [B,C], labels [B,H,W] uint8 -> dict(pred, hist, cand_idx, cand_score, threshold).
    images_u8 [B,16n,16n,3] uint8 (RGB): run the reference's DenseCRF on every visited candidate map."""
    C = sim.shape[2]
    num_classes = num_classes or C
    cand_idx, cand_score, thr = ops.seg_select(scores, top_cls_num, ncand)
    need_prob = refine is not None or images_u8 is not None
    masks, prob = ops.seg_masks(sim, cand_idx, num_patch, want_prob=need_prob)
    if need_prob:
        B, K, N = prob.shape
        lo = prob.view(B, K, num_patch, num_patch)
        if refine is not None:
            up = lo.repeat_interleave(16, 2).repeat_interleave(16, 3)
            masks = refine(up, cand_idx, cand_score).to(torch.uint8).contiguous()
        else:
            masks = crf_masks(lo, cand_idx, images_u8, scale=16)
    if closing:
        masks = ops.close7(masks, cand_idx.reshape(-1))                       # cv2.dilate then cv2.erode (:156-157), visited slots only
    pred, hist = ops.seg_predict(masks, cand_idx, cand_score, labels, num_classes, ignore_index, hist=hist, want_pred=want_pred)
    return {"pred": pred, "hist": hist, "cand_idx": cand_idx, "cand_score": cand_score, "threshold": thr, "masks": masks}


def eval_batch(model, image, label, text, top_cls_num, hist=None, crf=True, mean=None, std=None, sim_dtype=None, refine=None,
               want_pred=False):
    """One batch of the zero-shot segmentation evaluation, everything on the device (tools/seg_evaluation.py:99-170 for every image of
    the batch at once): towers -> pooled embedding + projected patch tokens -> similarity map for all classes -> segment().
    image [B,3,S,S] normalised network input, label [B,H,W] uint8, text [C,512] unit-norm class embeddings.  crf: run the DenseCRF on
    the de-normalised input (image * std + mean, :104) as the reference does; hist [3,C] int64 accumulates."""
    from .heads import patch_text_similarity
    feats = model.forward_image_feature(image)                    # [B, n*n, D]
    pooled = model.forward_image_project(feats)                   # [B, 512]
    n = int(round(feats.shape[1] ** 0.5))
    sim = patch_text_similarity(model.image_projection(feats), text, compute_dtype=sim_dtype)
    raw = None
    if crf and refine is None:
        raw = (((image * std) + mean) * 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
    return segment(sim, ops.gemm(pooled.float(), text), label, n, top_cls_num, hist=hist, want_pred=want_pred, refine=refine, images_u8=raw)


def iou_from_hist(hist):
    """hist [3,C] int64 -> (per-class IoU float64 with NaN where the class never occurs, mean over non-NaN) as :172-173."""
    inter, pred, label = hist[0].double(), hist[1].double(), hist[2].double()
    union = pred + label - inter
    iou = inter / union
    return iou, iou[~torch.isnan(iou)].mean()
