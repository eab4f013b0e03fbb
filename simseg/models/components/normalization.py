import torch
from torch.autograd import Function

from simseg_amd import ops

__all__ = ["L2norm"]


class _L2normRows(Function):
    """x / (||x||_2 + eps) over the last dim: a top-1 'pool' over a single token IS this op, so the fused
    pooling kernel provides forward and backward."""

    @staticmethod
    def forward(ctx, x, eps):
        rows = x.reshape(-1, 1, x.shape[-1]).contiguous().float()
        emb, idx, norm = ops.topk_pool_l2norm_fwd(rows, 1, None, eps)
        ctx.save_for_backward(emb, idx, norm)
        ctx.eps, ctx.shape = eps, x.shape
        return emb.view(x.shape)

    @staticmethod
    def backward(ctx, g):
        emb, idx, norm = ctx.saved_tensors
        d = ops.topk_pool_l2norm_bwd(g.reshape(emb.shape).contiguous().float(), emb, norm, idx, 1, torch.float32, ctx.eps)
        return d.view(ctx.shape), None


def L2norm(X, dim, eps=1e-8):
    """simseg/models/components/normalization.py:6-11"""
    if dim not in (-1, X.dim() - 1):
        X = X.transpose(dim, -1)
        return _L2normRows.apply(X, eps).transpose(dim, -1)
    if X.shape[-1] % 64 != 0 or X.shape[-1] > 1024:
        raise ValueError("L2norm kernel handles a last dim that is a multiple of 64 and <= 1024")
    return _L2normRows.apply(X, eps)
