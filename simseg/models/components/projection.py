import torch.nn as nn

from simseg_amd.nn import compute_dtype
from simseg_amd.towers import LinearFn

__all__ = ["SimpleProjection", "ComplexProjection"]


class SimpleProjection(nn.Module):
    """Per-token Linear(D -> projection_dim, bias=False): simseg/models/components/projection.py:29-46."""

    def __init__(self, cfg, embedding_dim, projection_dim, trainable=True):
        super().__init__()
        self.projection_dim = projection_dim
        self.linear = nn.Linear(embedding_dim, projection_dim, bias=False)
        nn.init.trunc_normal_(self.linear.weight, std=0.02)
        if not trainable:
            for p in self.linear.parameters():
                p.requires_grad = False

    def forward(self, x):
        return LinearFn.apply(x, self.linear.weight, None, compute_dtype())


class ComplexProjection(nn.Module):
    def __init__(self, *a, **kw):
        super().__init__()
        raise NotImplementedError("projection.name='complex' is not used by any shipped config and cannot be built by the "
                                  "reference either (its ctor rejects the `trainable` kwarg CLIPModel passes, clip.py:28-33)")
