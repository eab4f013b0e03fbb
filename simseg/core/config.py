"""Global config: code defaults < task defaults < YAML < `a.b.c=value` argv, then frozen
(mirror of the behaviour of simseg/core/config.py:13-309; implementation is our own).

  * unknown YAML key          -> KeyError('Non-existent config key: a.b')           (config.py:194-195)
  * unknown argv key          -> ValueError('Undefined attribute "x" detected ...') (config.py:164-165)
  * argv values: bare words become strings, literals are literal_eval'ed, lists may be written [a,b]
  * type coercion against the current value: str <- anything, list <-> tuple, "[a,b]" -> list; else ValueError
"""
import copy
import os
import re
from ast import literal_eval

import yaml

from simseg.utils.collections import AttrDict

__all__ = ["cfg", "update_cfg"]


def _to_attr(d):
    out = AttrDict()
    for k, v in d.items():
        out[k] = _to_attr(v) if isinstance(v, dict) and v.pop("__attr__", True) else v
    return out


def _plain(**kw):
    """A dict-valued LEAF (cfg.optim.param etc. are plain dicts in the reference, replaced wholesale by YAML)."""
    kw["__attr__"] = False
    return kw


def _core_defaults():
    return dict(
        epoch=None, seed=None, mae_seed=False, inference=False,
        runner=dict(name=None, val_interval=1, val_interval_steps=-1),
        dist=dict(name="apex", enable_adasum=False, enable_adascale=False, fp16=False, param=_plain()),
        model=dict(name=None),
        data=dict(name=None, batch_size=None, batch_size_val=None, train_steps=None, val_steps=None),
        optim=dict(name="SGD", param=_plain(momentum=0.9, weight_decay=1e-4), param_group_rules=_plain(), grad_clip=_plain(),
                   lr=dict(name="constant_schedule", init=0.01, warmup_proportion=0.1, warmup_epoch=None, param=_plain())),
        ckpt=dict(dir=None, step_interval=500, filename="latest_ckpt.pth", external_resume=None, auto_resume=True, soft_resume=False),
        log=dict(interval_train=100, interval_val=100),
    )


cfg = _to_attr(_core_defaults())


def _reset():
    cfg.set_this_dict_immutable(False)
    cfg.clear()
    cfg.update(_to_attr(_core_defaults()))


def _merge(src, dst, path=()):
    for k, v in src.items():
        here = path + (k,)
        if k not in dst:
            raise KeyError("Non-existent config key: {}".format(".".join(here)))
        if isinstance(dst[k], AttrDict):
            if not isinstance(v, dict):
                raise AssertionError(f"value for {'.'.join(here)} must be a dict, got {v} instead")
            _merge(v, dst[k], here)
        else:
            dst[k] = copy.deepcopy(v)


_WORD = re.compile(r"[^\[\]{},\s:]+|[^\[\]{},\s]+")


def _looks_literal(tok):
    if tok in ("True", "False", "None"):
        return True
    try:
        float(tok)
        return True
    except ValueError:
        return (tok[0] == tok[-1] and tok[0] in "'\"" and len(tok) >= 2)


def _quote_bare_words(s):
    """'[f30k,coco]' -> '["f30k","coco"]'; numbers / booleans / None / quoted strings stay as they are.
    ':' separates tokens only inside a {...} literal."""
    seps = "[]{}, " + (":" if "{" in s and "}" in s else "")
    out, tok = [], ""
    for ch in s:
        if ch in seps:
            if tok:
                out.append(tok if _looks_literal(tok) else f'"{tok}"')
                tok = ""
            out.append(ch)
        else:
            tok += ch
    if tok:
        out.append(tok if _looks_literal(tok) else f'"{tok}"')
    return "".join(out)


def _decode(text):
    try:
        return literal_eval(text)
    except (ValueError, SyntaxError):
        return text


def _coerce(new, old, key):
    if old is None or type(new) is type(old):
        return new
    if isinstance(old, str):
        return str(new)
    if isinstance(new, tuple) and isinstance(old, list):
        return list(new)
    if isinstance(new, list) and isinstance(old, tuple):
        return tuple(new)
    if isinstance(new, str) and isinstance(old, list):
        return (new[1:-1] if new.startswith("[") and new.endswith("]") else new).split(",")
    raise ValueError("Type mismatch ({} vs. {}) with values ({} vs. {}) for config key: {}".format(type(old), type(new), old, new, key))


def _apply_argv(argv):
    for item in argv:
        if "=" not in item:
            raise AssertionError("Error argv (must be key=value): " + item)
        key, raw = item.split("=", 1)
        node, trail = cfg, "cfg"
        parts = key.split(".")
        for i, part in enumerate(parts):
            if not isinstance(node, dict) or part not in node:
                raise ValueError(f'Undefined attribute "{part}" detected for "{trail}"')
            if i < len(parts) - 1:
                node, trail = node[part], f"{trail}.{part}"
        value = _decode(_quote_bare_words(raw))
        if isinstance(value, dict):
            value = AttrDict(value)
        node[parts[-1]] = _coerce(value, node[parts[-1]], key)


def update_cfg(task_cfg_init_fn, cfg_yaml, cfg_argv, preprocess_fn=None):
    """Returns the (global, frozen) cfg built from task defaults, the YAML file and `key=value` overrides."""
    _reset()
    task_cfg_init_fn(cfg)
    if not os.path.exists(cfg_yaml):
        raise ValueError(f"cfg file not found: {cfg_yaml}")
    with open(cfg_yaml) as f:
        _merge(yaml.load(f, Loader=yaml.FullLoader) or {}, cfg)
    _apply_argv(cfg_argv or [])
    if preprocess_fn:
        preprocess_fn(cfg)
    cfg.set_this_dict_immutable(True)
    return cfg
