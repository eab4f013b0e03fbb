"""State-dict key helpers (mirror of simseg/utils/checkpoint_utils.py:4-36)."""
from collections import OrderedDict

__all__ = ["filter_state", "convert_keys"]


def filter_state(state_dict, remove_prefixes=(), keep_prefixes=None):
    out = OrderedDict()
    for k, v in state_dict.items():
        if any(k.startswith(p) for p in remove_prefixes):
            continue
        if keep_prefixes is not None and not any(k.startswith(p) for p in keep_prefixes):
            continue
        out[k] = v
    return out


def convert_keys(state_dict, change_list):
    """change_list: [[old_prefix, new_prefix], ...]"""
    out = OrderedDict()
    for k, v in state_dict.items():
        for old, new in change_list or ():
            if k.startswith(old):
                k = new + k[len(old):]
                break
        out[k] = v
    return out
