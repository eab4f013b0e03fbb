"""Attribute-style nested config dict with a recursive freeze switch (mirror of simseg/utils/collections.py:8-49)."""

__all__ = ["AttrDict"]

_FROZEN = "__immutable__"


class AttrDict(dict):
    IMMUTABLE = _FROZEN

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        object.__setattr__(self, _FROZEN, False)

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError:
            raise AttributeError(key) from None

    def __setattr__(self, key, value):
        if self.__dict__[_FROZEN]:
            raise AttributeError(f'Attempted to set "{key}" to "{value}", but AttrDict is immutable')
        if key in self.__dict__:
            object.__setattr__(self, key, value)
        else:
            self[key] = value

    def set_this_dict_immutable(self, is_immutable):
        object.__setattr__(self, _FROZEN, bool(is_immutable))
        for child in self.values():
            if isinstance(child, AttrDict):
                child.set_this_dict_immutable(is_immutable)

    def is_this_dict_immutable(self):
        return self.__dict__[_FROZEN]
