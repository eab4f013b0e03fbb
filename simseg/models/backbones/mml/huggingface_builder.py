"""`huggingface_modelzoo` backbone (mirror of simseg/models/backbones/mml/huggingface_builder.py:6-23): a BERT encoder
without pooler whose output exposes `.last_hidden_state` -- here the MI355X-native tower."""
import torch.nn as nn

from simseg_amd.nn import Bert

from ..builder import BACKBONE
from ._weights import maybe_load_pretrained

__all__ = ["HuggingFaceModel", "huggingface_modelzoo"]


class HuggingFaceModel(nn.Module):
    def __init__(self, cfg, **kwargs):
        super().__init__()
        tag = cfg.model.text_encoder.tag
        self.model = Bert(tag)
        if cfg.model.text_encoder.pretrained:
            maybe_load_pretrained(self.model, tag)

    def forward(self, input_ids, attention_mask, **kwargs):
        return self.model(input_ids=input_ids, attention_mask=attention_mask)


@BACKBONE.register_obj
def huggingface_modelzoo(cfg, **kwargs):
    return HuggingFaceModel(cfg, **kwargs)
