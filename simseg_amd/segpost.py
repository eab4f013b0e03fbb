"""Per-image body of the reference's `evaluate_benchmark` (tools/seg_evaluation.py:99-170) as batched device work.

    feats  = model.forward_image_feature(image)              # [B, n*n, D]
    pooled = model.forward_image_project(feats)              # [B, 512]
    sim    = patch_text_similarity(model.image_projection(feats), text)      # [B, n*n, C]   (K14)
    out    = segment(sim, pooled @ text.T, labels, n, top_cls_num)

The reference runs a CPU DenseCRF (pydensecrf, absent from this image) between the normalised map and the morphology; here
the binary map is the CRF's unary decision (prob > 0.5).  A caller that has a CRF passes `refine`: it receives the
probabilities [B,ncand,16n,16n] and the candidate table and returns uint8 masks of the same shape (0/255) - everything
after it (7x7 dilate + erode, nearest resize, score-weighted argmax, IoU histograms) is the reference's arithmetic again.
"""
import torch

from . import ops


def segment(sim, scores, labels, num_patch, top_cls_num, num_classes=None, ncand=5, ignore_index=255, refine=None, hist=None,
            want_pred=True, closing=True):
    """sim [B,n*n,C] fp32, scores [B,C], labels [B,H,W] uint8 -> dict(pred, hist, cand_idx, cand_score, threshold)."""
    C = sim.shape[2]
    num_classes = num_classes or C
    cand_idx, cand_score, thr = ops.seg_select(scores, top_cls_num, ncand)
    masks, prob = ops.seg_masks(sim, cand_idx, num_patch, want_prob=refine is not None)
    if refine is not None:
        B, K, N = prob.shape
        up = prob.view(B, K, num_patch, num_patch).repeat_interleave(16, 2).repeat_interleave(16, 3)
        masks = refine(up, cand_idx, cand_score).to(torch.uint8).contiguous()
    if closing:
        masks = ops.close7(masks, cand_idx.reshape(-1))                       # cv2.dilate then cv2.erode (:156-157), visited slots only
    pred, hist = ops.seg_predict(masks, cand_idx, cand_score, labels, num_classes, ignore_index, hist=hist, want_pred=want_pred)
    return {"pred": pred, "hist": hist, "cand_idx": cand_idx, "cand_score": cand_score, "threshold": thr, "masks": masks}


def iou_from_hist(hist):
    """hist [3,C] int64 -> (per-class IoU float64 with NaN where the class never occurs, mean over non-NaN) as :172-173."""
    inter, pred, label = hist[0].double(), hist[1].double(), hist[2].double()
    union = pred + label - inter
    iou = inter / union
    return iou, iou[~torch.isnan(iou)].mean()
