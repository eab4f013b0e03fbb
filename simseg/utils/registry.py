"""Name -> callable registries (mirror of simseg/utils/registry.py:7-81)."""

__all__ = ["Registry", "build_from_cfg"]


class Registry:
    def __init__(self, name):
        self._name = name
        self._items = {}

    name = property(lambda self: self._name)
    obj_dict = property(lambda self: self._items)

    def __repr__(self):
        return f"Registry(name={self._name}, items={list(self._items)})"

    def get(self, key):
        return self._items.get(key)

    def has(self, key):
        return key in self._items

    def register_obj(self, obj):
        if not callable(obj):
            raise TypeError(f"object {obj} must be callable")
        if obj.__name__ in self._items:
            raise KeyError(f"{obj.__name__} is already registered in {self._name}.")
        self._items[obj.__name__] = obj
        return obj


def build_from_cfg(name, cfg, registry, default_args=None):
    if default_args is not None and not isinstance(default_args, dict):
        raise AssertionError("default_args must be a dict or None")
    factory = registry.get(name)
    if factory is None:
        raise KeyError(f"{name} is not in the {registry.name} registry. Choose among {list(registry.obj_dict)}")
    for k, v in (default_args or {}).items():
        cfg.setdefault(k, v)
    return factory(cfg)
