"""Prompt templates for the zero-shot classifier (the role of simseg/utils/prompt.py).  The reference ships OpenAI's
80 ImageNet templates as a source file; this repo does not copy that list.  `openai_imagenet_template` reads one
template per line (with `{}` for the class name) from $SIMSEG_PROMPT_TEMPLATES or data/prompt_templates.txt if present
-- drop the reference's list there for checkpoint-faithful numbers -- and otherwise falls back to a short generic set."""
import os

__all__ = ["openai_imagenet_template", "load_templates"]

_FALLBACK = ("a photo of a {}.", "a photo of the {}.", "a cropped photo of a {}.", "a close-up photo of a {}.",
             "a bright photo of a {}.", "a dark photo of a {}.", "a photo of a large {}.", "a photo of a small {}.",
             "a blurry photo of a {}.", "a good photo of a {}.", "a drawing of a {}.", "a painting of a {}.",
             "a rendering of a {}.", "itap of a {}.", "there is a {} in the scene.", "this is a {}.")
_cache = None


def load_templates():
    global _cache
    if _cache is None:
        here = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        for path in (os.environ.get("SIMSEG_PROMPT_TEMPLATES"), os.path.join(here, "data", "prompt_templates.txt")):
            if path and os.path.exists(path):
                with open(path) as f:
                    lines = [ln.strip() for ln in f if "{}" in ln]
                if lines:
                    _cache = tuple(lines)
                    break
        else:
            _cache = _FALLBACK
    return _cache


def openai_imagenet_template(classname):
    return [t.format(classname) for t in load_templates()]
