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


def _pil_resample_restated(a, H, W, kind):
    """Pillow's two-pass resampling (ImagingResample: horizontal then vertical, uint8 between the passes; filter support scaled by the
    reduction factor = its built-in antialiasing) restated with explicit loops: what torchvision's Resize on a PIL image computes
    (transforms/mml/transforms.py:15-22 -> F.resize -> Image.resize).  triangle filter = BILINEAR, Keys cubic a = -0.5 = BICUBIC."""
    def tri(x):
        x = abs(x)
        return 1 - x if x < 1 else 0.0

    def cubic(x, A=-0.5):
        x = abs(x)
        if x < 1:
            return ((A + 2) * x - (A + 3)) * x * x + 1
        if x < 2:
            return (((x - 5) * x + 8) * x - 4) * A
        return 0.0

    flt, sup = (tri, 1.0) if kind == "bilinear" else (cubic, 2.0)

    def one_axis(arr, out_size, axis):
        in_size = arr.shape[axis]
        scale = in_size / out_size
        fs = max(scale, 1.0)
        support = sup * fs
        arr = np.moveaxis(arr, axis, 0).astype(np.float64)
        out = np.zeros((out_size,) + arr.shape[1:])
        for i in range(out_size):
            c = (i + 0.5) * scale
            lo, hi = max(int(c - support + 0.5), 0), min(int(c + support + 0.5), in_size)
            w = np.array([flt((j + 0.5 - c) / fs) for j in range(lo, hi)])
            out[i] = np.tensordot(w / w.sum(), arr[lo:hi], 1)
        return np.clip(np.rint(np.moveaxis(out, 0, axis)), 0, 255)

    return one_axis(one_axis(a, W, 1), H, 0).astype(np.uint8)


def test_resize_interpolation_arithmetic_and_crop_geometry():
    """The interpolation arithmetic of the valid transforms on a NON-constant image (round 4's test used a constant one): `resize` (square,
    bilinear, down- and up-sampling), `resize_bicubic` (shorter side -> size, aspect kept, rounding of the longer side as torchvision
    computes it) and `center_crop` (torchvision's rounding of the offsets), against the restated Pillow resampler - within 1 grey level
    (Pillow accumulates in fixed point, the restatement in float64)."""
    from simseg.transforms import TRANSFORMS
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    img = Image.fromarray(a)
    for size in (24, 64):
        out = np.asarray(TRANSFORMS.get("resize")(_cfg([f"transforms.resize.size={size}"]))(img))
        ref = _pil_resample_restated(a, size, size, "bilinear")
        assert out.shape == (size, size, 3) and int(np.abs(out.astype(int) - ref.astype(int)).max()) <= 1
        assert float((out != ref).mean()) < 0.05
    for size, hw in ((24, (24, 34)), (74, (74, 106))):                   # 37 x 53: the shorter side is the height; 53 * size / 37 rounded
        out = np.asarray(TRANSFORMS.get("resize_bicubic")(_cfg([f"transforms.resize_bicubic.size={size}"]))(img))
        assert out.shape[:2] == hw
        ref = _pil_resample_restated(a, hw[0], hw[1], "bicubic")
        assert int(np.abs(out.astype(int) - ref.astype(int)).max()) <= 1
    tall = Image.fromarray(np.ascontiguousarray(a.transpose(1, 0, 2)))    # 53 x 37: now the width is the shorter side
    assert np.asarray(TRANSFORMS.get("resize_bicubic")(_cfg(["transforms.resize_bicubic.size=24"]))(tall)).shape[:2] == (34, 24)
    crop = np.asarray(TRANSFORMS.get("center_crop")(_cfg(["transforms.center_crop.size=20"]))(img))
    top, left = int(round((37 - 20) / 2.0)), int(round((53 - 20) / 2.0))
    np.testing.assert_array_equal(crop, a[top:top + 20, left:left + 20])


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
