"""Process-wide environment singleton `ENV` (mirror of simseg/utils/context.py:13-119)."""
from functools import wraps

import torch

from .collections import AttrDict

__all__ = ["ENV"]


def _checked(name, check):
    priv = "_" + name

    def getter(self):
        return getattr(self, priv)

    def setter(self, value):
        assert check(value), f"bad value for ENV.{name}: {value!r}"
        setattr(self, priv, value)

    return property(getter, setter)


class GlobalContext:
    _instance = None
    _cfg, _dist_mode, _rank, _size, _local_rank, _device, _loader_type = None, None, 0, 1, 0, 0, None

    def __new__(cls):
        if cls._instance is None:
            cls._instance = super().__new__(cls)
        return cls._instance

    rank = _checked("rank", lambda v: isinstance(v, int) and v >= 0)
    size = _checked("size", lambda v: isinstance(v, int) and v >= 0)
    local_rank = _checked("local_rank", lambda v: isinstance(v, int) and v >= 0)
    device = _checked("device", lambda v: isinstance(v, torch.device))
    dist_mode = _checked("dist_mode", lambda v: v in ("apex", "horovod", "torch", None))
    loader_type = _checked("loader_type", lambda v: v in ("local", "parquet"))
    cfg = _checked("cfg", lambda v: isinstance(v, AttrDict))

    def _only(self, attr):
        def deco(func):
            @wraps(func)
            def wrapper(*a, **kw):
                if getattr(self, attr) == 0:
                    return func(*a, **kw)
            return wrapper
        return deco

    def root_only(self, func):
        return self._only("_rank")(func)

    def local_root_only(self, func):
        return self._only("_local_rank")(func)

    @classmethod
    def cls_root_only(cls, func):
        return cls._instance._only("_rank")(func) if cls._instance else func


ENV = GlobalContext()
