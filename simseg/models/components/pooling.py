import torch
import torch.nn as nn
from torch.autograd import Function

from simseg_amd import ops

__all__ = ["TopKPooling", "AvgPooling", "clip_k_to_shortest"]


def clip_k_to_shortest(k, attention_mask):
    """pooling.py:61-63: k never exceeds the shortest caption of the batch.  k == 1 needs no look at the mask
    (every caption has its [CLS]); larger k costs the same host sync the reference pays."""
    if attention_mask is None or k <= 1:
        return k
    return max(1, min(k, int(attention_mask.sum(1).min())))


class _TopKPool(Function):
    @staticmethod
    def forward(ctx, x, k, mask):
        x = x.contiguous()
        pooled, idx, norm = ops.topk_pool_l2norm_fwd(x, k, mask, normalize=False)
        ctx.save_for_backward(pooled, idx, norm)
        ctx.N, ctx.dtype = x.shape[1], x.dtype
        return pooled if x.dtype == torch.float32 else pooled

    @staticmethod
    def backward(ctx, g):
        pooled, idx, norm = ctx.saved_tensors
        d = ops.topk_pool_l2norm_bwd(g.contiguous().float(), pooled, norm, idx, ctx.N, ctx.dtype, normalize=False)
        return d, None, None


class TopKPooling(nn.Module):
    """LoDA pooling (simseg/models/components/pooling.py:42-65): per (batch, channel) mean of the k largest token
    values; masked tokens count as -10000.  The reference overwrites its input in place; this one does not."""

    def __init__(self, k, dim):
        super().__init__()
        if dim != 1:
            raise NotImplementedError("TopKPooling pools over the token dim (dim=1)")
        self.k, self.dim = k, dim

    def forward(self, x, attention_mask=None):
        k = clip_k_to_shortest(self.k, attention_mask)
        mask = None if attention_mask is None else attention_mask.contiguous().long()
        return _TopKPool.apply(x if x.dtype in (torch.float32, torch.bfloat16, torch.float16) else x.float(), k, mask)


class AvgPooling(nn.Module):
    """pooling.py:7-19 (cfg.model.pool.name == 'avg'; not selected by the shipped YAMLs)."""

    def forward(self, x, attention_mask=None):
        raise NotImplementedError("pool.name='avg' is outside the accelerated path (SURVEY.md 2.1 row 3); use 'loda'")
