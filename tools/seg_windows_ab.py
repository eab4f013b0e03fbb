#!/usr/bin/env python
"""Seg-eval legs of bench.py at two batch sizes: 63 / 64 windows (the tile-grid fit of round 2) against 256 (every GEMM row count a
multiple of 256: full tiles only, so the persistent ping-pong kernel takes them)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
dev = torch.device("cuda", 0)
vs = dict(img=288, classes=21, tag="vit_small_patch16_224_in21k", dim=384)
for name, dtype, crf, kw, ws in (("ViT-B@512 bf16 + CRF", "bf16", True, {}, (63, 256)), ("ViT-S@288 bf16", "bf16", False, vs, (64, 256)),
                                  ("ViT-S@288 fp32", "fp32", False, vs, (64, 256)), ("ViT-S@288 fp32 + CRF", "fp32", True, vs, (64, 256))):
    for w in ws:
        r = bench.seg_eval_bench(dev, 1, dtype, crf=crf, steps=1 if crf else 2, windows=w, **kw)
        print(f"{name}: {w} windows per batch -> {r['windows_per_s']} windows/s", flush=True)
