"""Small helpers the path uses (mirror of the used part of simseg/utils/misc.py)."""
import torch

__all__ = ["calc_topk_accuracy", "AverageMeter", "is_number_or_bool_or_none"]


def calc_topk_accuracy(output, target, topk=(1,)):
    """misc.py:462-478: fraction of rows whose target is among the k largest scores."""
    maxk = max(topk)
    pred = output.topk(maxk, 1, True, True)[1]
    hit = pred == target.view(-1, 1)
    n = torch.sum(target >= 0)
    return [hit[:, :k].any(dim=1).float().sum() / n for k in topk]


class AverageMeter:
    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def is_number_or_bool_or_none(x):
    try:
        float(x)
        return True
    except (TypeError, ValueError):
        return x in ("True", "False", "None")
