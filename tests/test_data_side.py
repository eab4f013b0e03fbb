"""CPU tests of the data-side import paths the reference's eval tools use, on tiny synthetic datasets."""
import io
import os

import numpy as np
import pytest
import torch
from PIL import Image

from conftest import REPO


def _cfg(argv):
    from simseg.core.config import update_cfg
    from simseg.tasks.clip.config import task_cfg_init_fn, update_clip_config
    return update_cfg(task_cfg_init_fn, os.path.join(REPO, "configs/clip/simseg.vit-s.yaml"), argv, update_clip_config)


def test_valid_transforms_resize_normalize():
    from simseg.transforms import build_transforms
    cfg = _cfg(["transforms.resize.size=32"])
    t = build_transforms(cfg, mode="valid")
    img = Image.fromarray(np.full((50, 70, 3), 128, dtype=np.uint8))
    x = t(img)
    assert x.shape == (3, 32, 32) and x.dtype == torch.float32
    mean, std = torch.tensor(cfg.transforms.normalize.mean), torch.tensor(cfg.transforms.normalize.std)
    torch.testing.assert_close(x[:, 0, 0], (128 / 255.0 - mean) / std)
    with pytest.raises(NotImplementedError):
        build_transforms(cfg, mode="train")        # autoaug etc. are outside the path


def test_seg_loader_layouts(tmp_path):
    from simseg.datasets.seg.seg_dataset import build_torch_valid_loader
    root = tmp_path / "VOCdevkit" / "VOC2012"
    for d in ("JPEGImages", "SegmentationClass", "ImageSets/Segmentation"):
        (root / d).mkdir(parents=True)
    names = ["a", "b", "c"]
    (root / "ImageSets/Segmentation/val.txt").write_text("\n".join(names) + "\n")
    rng = np.random.RandomState(0)
    for n in names:
        Image.fromarray(rng.randint(0, 255, (40, 60, 3), dtype=np.uint8)).save(root / "JPEGImages" / f"{n}.jpg")
        Image.fromarray(rng.randint(0, 21, (40, 60), dtype=np.uint8)).save(root / "SegmentationClass" / f"{n}.png")
    coco = tmp_path / "coco_stuff164k"
    (coco / "images/val2017").mkdir(parents=True); (coco / "annotations/val2017").mkdir(parents=True)
    Image.fromarray(rng.randint(0, 255, (20, 20, 3), dtype=np.uint8)).save(coco / "images/val2017/000001.jpg")
    Image.fromarray(rng.randint(0, 80, (20, 20), dtype=np.uint8)).save(coco / "annotations/val2017/000001_labelTrainIds.png")
    cfg = _cfg([f"data.data_path={tmp_path}", "data.num_workers=0", "transforms.resize.size=32"])
    loader = build_torch_valid_loader(cfg, "pascal_voc")
    batches = list(loader)
    assert len(batches) == 3
    img, lab = batches[0]
    assert img.shape == (1, 3, 32, 32) and lab.shape == (1, 40, 60) and lab.dtype == torch.uint8
    img, lab = next(iter(build_torch_valid_loader(cfg, "coco_stuff")))
    assert img.shape == (1, 3, 32, 32) and lab.shape == (1, 20, 20)
    with pytest.raises(NotImplementedError):
        build_torch_valid_loader(cfg, "ade20k")


def test_parquet_valid_loader(tmp_path):
    import pandas as pd
    from simseg.datasets.clip.clip_dataset import build_parquet_valid_loader
    rng = np.random.RandomState(1)
    rows = []
    for i in range(6):
        buf = io.BytesIO()
        Image.fromarray(rng.randint(0, 255, (30, 30, 3), dtype=np.uint8)).save(buf, format="JPEG")
        rows.append(dict(imbytes=buf.getvalue(), caption=f"caption number {i}", image_id=i // 2, id=i))
    (tmp_path / "coco").mkdir()
    pd.DataFrame(rows).to_parquet(tmp_path / "coco" / "valid.parquet")

    class Tok:           # stand-in with the HF call signature (no vocab files offline)
        def __call__(self, text, padding, truncation, max_length):
            ids = [101] + [1000 + (hash(w) % 500) for w in text.split()][: max_length - 2] + [102]
            mask = [1] * len(ids) + [0] * (max_length - len(ids))
            return {"input_ids": ids + [0] * (max_length - len(ids)), "attention_mask": mask}

    cfg = _cfg([f"data.data_path={tmp_path}", "data.num_workers=0", "data.batch_size_val=4", "transforms.resize.size=32"])
    loader = build_parquet_valid_loader(cfg, "coco", tokenizer=Tok())
    image, ids, mask, caption, image_id, caption_id = next(iter(loader))
    assert image.shape == (4, 3, 32, 32) and ids.shape == (4, 25) and mask.shape == (4, 25)
    assert list(caption_id) == [0, 1, 2, 3] and list(image_id) == [0, 0, 1, 1] and caption[2] == "caption number 2"
    assert int(mask[0].sum()) == 5 and int(ids[0, 0]) == 101
