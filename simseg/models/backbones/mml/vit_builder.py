"""`vit_modelzoo` backbone (mirror of simseg/models/backbones/mml/vit_builder.py:8-27): the timm ViT feature extractor
returning ALL tokens [B, 1+N, D] after the final LayerNorm -- here the MI355X-native tower."""
import torch.nn as nn

from simseg_amd.nn import ViT

from ..builder import BACKBONE
from ._weights import maybe_load_pretrained

__all__ = ["ViTModel", "vit_modelzoo"]


class ViTModel(nn.Module):
    def __init__(self, cfg, img_size=224, **kwargs):
        super().__init__()
        tag = cfg.model.image_encoder.tag
        self.model = ViT(tag, img_size=img_size)
        if cfg.model.image_encoder.pretrained:
            maybe_load_pretrained(self.model, tag)

    def forward(self, x):
        return self.model(x)


@BACKBONE.register_obj
def vit_modelzoo(cfg, **kwargs):
    return ViTModel(cfg, **kwargs)
