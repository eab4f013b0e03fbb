"""Validation loaders for the zero-shot segmentation benchmarks with the reference's directory layouts and call surface
(simseg/datasets/seg/seg_dataset.py:13-81): PASCAL VOC 2012, PASCAL Context, COCO-Stuff-164k (trainId PNGs)."""
import os
from glob import glob

import numpy as np
import torch
from PIL import Image

from simseg.datasets.builder import DATALOADER
from simseg.transforms import build_transforms

__all__ = ["SegDataset", "build_torch_valid_loader", "seg"]

_LAYOUT = {
    # name: (root parts, image dir, label dir, list file or None, label suffix)
    "pascal_voc": (("VOCdevkit", "VOC2012"), ("JPEGImages",), ("SegmentationClass",), ("ImageSets", "Segmentation", "val.txt"), ""),
    "pascal_context": (("VOCdevkit", "VOC2010"), ("JPEGImages",), ("SegmentationClassContext",), ("ImageSets", "SegmentationContext", "val.txt"), ""),
    "coco_stuff": (("coco_stuff164k",), ("images", "val2017"), ("annotations", "val2017"), None, "_labelTrainIds"),
}


class SegDataset(torch.utils.data.Dataset):
    def __init__(self, cfg, dataset_name, data_path, transforms=None):
        if dataset_name not in _LAYOUT:
            raise NotImplementedError("Please verify dataset name.")
        root, img, lab, lst, self.suffix = _LAYOUT[dataset_name]
        self.cfg, self.name, self.transforms = cfg, dataset_name, transforms
        self.root_path = os.path.join(data_path, *root)
        self.image_path = os.path.join(self.root_path, *img)
        self.label_path = os.path.join(self.root_path, *lab)
        if lst is not None:
            list_path = os.path.join(self.root_path, *lst)
            if not os.path.exists(list_path):
                raise AssertionError(f"missing split file {list_path}")
            with open(list_path) as f:
                self.name_list = [ln.rstrip() for ln in f if ln.strip()]
        else:
            self.name_list = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob(os.path.join(self.image_path, "*.jpg")))
        self.length = len(self.name_list)

    def __getitem__(self, index):
        item = self.name_list[index]
        image = Image.open(os.path.join(self.image_path, item) + ".jpg").convert("RGB")
        if self.transforms is not None:
            image = self.transforms(image)
        label = torch.tensor(np.array(Image.open(os.path.join(self.label_path, item + self.suffix) + ".png")))
        return image, label

    def __len__(self):
        return self.length


def build_torch_valid_loader(cfg, name, mode="valid", **kwargs):
    """Not distributed, like the reference (every rank sees the whole set, seg_dataset.py:74-80)."""
    ds = SegDataset(cfg=cfg, dataset_name=name, data_path=cfg.data.data_path, transforms=build_transforms(cfg, mode=mode))
    return torch.utils.data.DataLoader(ds, batch_size=cfg.data.batch_size_val, num_workers=cfg.data.num_workers, pin_memory=True,
                                       drop_last=False)


@DATALOADER.register_obj
def seg(cfg):
    valid = [build_torch_valid_loader(cfg, name, mode="valid") for name in cfg.data.valid_name] if cfg.data.enable_valid else []
    return None, valid
