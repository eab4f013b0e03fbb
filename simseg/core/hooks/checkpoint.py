"""Only the piece of simseg/core/hooks/checkpoint.py the eval tools import (tools/seg_evaluation.py:233)."""
from collections import OrderedDict

from simseg.utils import ENV


def get_dist_state_dict(state_dict):
    """checkpoint.py:48-56: checkpoints store the unwrapped model; under torch/apex DDP the keys need `module.`."""
    if ENV.dist_mode in ("torch", "apex"):
        return OrderedDict((k if k.startswith("module.") else "module." + k, v) for k, v in state_dict.items())
    return state_dict
