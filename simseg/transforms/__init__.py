"""Eval-time image transforms with the reference's builder API (simseg/transforms/mml/transforms.py:74-93) but without
torchvision (absent here): PIL resampling + torch tensors.  Training-time augmentations (autoaug, random_resize_crop,
color ops, random erasing) are host-side data prep outside the accelerated path and are not provided."""
import numpy as np
import torch
from PIL import Image

from simseg.utils import logger
from simseg.utils.registry import Registry

__all__ = ["TRANSFORMS", "build_transforms"]

TRANSFORMS = Registry("TRANSFORMS")


class Compose:
    def __init__(self, ops):
        self.ops = ops

    def __call__(self, x):
        for op in self.ops:
            x = op(x)
        return x

    def __repr__(self):
        return "Compose(" + ", ".join(getattr(o, "__name__", o.__class__.__name__) for o in self.ops) + ")"


@TRANSFORMS.register_obj
def resize(cfg, **kwargs):
    size = cfg.transforms.resize.size

    def resize_op(img):                       # torchvision Resize((s, s)) on PIL: bilinear
        return img.resize((size, size), Image.BILINEAR)
    return resize_op


@TRANSFORMS.register_obj
def resize_bicubic(cfg, **kwargs):
    size = cfg.transforms.resize_bicubic.size

    def resize_bicubic_op(img):               # Resize(size, interpolation=BICUBIC): shorter side -> size
        w, h = img.size
        if w <= h:
            return img.resize((size, max(1, round(h * size / w))), Image.BICUBIC)
        return img.resize((max(1, round(w * size / h)), size), Image.BICUBIC)
    return resize_bicubic_op


@TRANSFORMS.register_obj
def center_crop(cfg, **kwargs):
    size = cfg.transforms.center_crop.size

    def center_crop_op(img):
        w, h = img.size
        left, top = int(round((w - size) / 2.0)), int(round((h - size) / 2.0))
        return img.crop((left, top, left + size, top + size))
    return center_crop_op


def _to_tensor(img):
    a = np.asarray(img, dtype=np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    return torch.from_numpy(a.copy()).permute(2, 0, 1).float().div_(255.0)


@TRANSFORMS.register_obj
def normalize(cfg, **kwargs):
    mean = torch.tensor(cfg.transforms.normalize.mean).view(-1, 1, 1)
    std = torch.tensor(cfg.transforms.normalize.std).view(-1, 1, 1)

    def normalize_op(t):
        return (t - mean) / std
    return normalize_op


def build_transforms(cfg, mode="train"):
    names = cfg.transforms.train_transforms if mode == "train" else cfg.transforms.valid_transforms
    ops = []
    for name in names:
        factory = TRANSFORMS.get(name)
        if factory is None:
            raise NotImplementedError(f"transform {name!r} is a training-time augmentation outside the accelerated path; "
                                      f"available: {sorted(TRANSFORMS.obj_dict)}")
        ops.append(factory(cfg))
    ops.extend([_to_tensor, TRANSFORMS.get("normalize")(cfg)])
    t = Compose(ops)
    logger.emph(f"{mode} image transform is composed of:", t)
    return t
