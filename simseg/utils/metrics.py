"""Per-class intersection / union histograms for mIoU (mirror of simseg/utils/metrics.py:5-99).  Host-side, after the
CPU CRF stage of the seg tool -- outside the accelerated path; kept in torch."""
import numpy as np
import torch

__all__ = ["intersect_and_union", "mean_iou"]


def _as_tensor(x):
    return torch.from_numpy(x) if isinstance(x, np.ndarray) else x


def intersect_and_union(pred_label, label, num_classes, ignore_index, label_map=None, reduce_zero_label=False):
    pred_label, label = _as_tensor(pred_label).long(), _as_tensor(label).long()
    if label_map:
        for old, new in label_map.items():
            label[label == old] = new
    if reduce_zero_label:
        label[label == 0] = 255
        label = label - 1
        label[label == 254] = 255
    keep = label != ignore_index
    pred_label, label = pred_label[keep], label[keep]
    inter = pred_label[pred_label == label]

    def hist(t):
        return torch.histc(t.float(), bins=num_classes, min=0, max=num_classes - 1)

    area_i, area_p, area_l = hist(inter), hist(pred_label), hist(label)
    return area_i, area_p + area_l - area_i, area_p, area_l


def mean_iou(results, gt_seg_maps, num_classes, ignore_index, nan_to_num=None, label_map=None, reduce_zero_label=False):
    """Sums the histograms over the images and returns (total_intersect, total_union) as float64 tensors, which is
    what tools/seg_evaluation.py:164-172 accumulates."""
    ti = torch.zeros(num_classes, dtype=torch.float64)
    tu = torch.zeros(num_classes, dtype=torch.float64)
    for pred, gt in zip(results, gt_seg_maps):
        i, u, _, _ = intersect_and_union(pred, gt, num_classes, ignore_index, label_map or {}, reduce_zero_label)
        ti += i.double()
        tu += u.double()
    return ti, tu
