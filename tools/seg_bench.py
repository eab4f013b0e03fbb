#!/usr/bin/env python
"""The seg-eval legs of bench.py alone (quick iteration on the eval path).  python tools/seg_bench.py [crf|nocrf] [chunk]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

crf = (sys.argv[1] if len(sys.argv) > 1 else "crf") == "crf"
dev = torch.device("cuda", 0)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
legs = (("bf16", {}), ("fp32", {}), ("fp32", dict(windows=64, img=288, classes=21, tag="vit_small_patch16_224_in21k", dim=384)))
sel = os.environ.get("SEG_BENCH_LEGS")           # e.g. "1" = the fp32 ViT-B leg only (profiles)
for dtype, kw in (legs if not sel else [legs[int(i)] for i in sel.split(",")]):
    r = bench.seg_eval_bench(dev, 1, dtype, crf=crf, steps=int(os.environ.get("SEG_BENCH_STEPS", "3")), **kw)
    print(json.dumps({k: r[k] for k in ("dtype", "window", "dense_crf", "windows_per_s", "post_ms_per_step", "post_visited_candidates_per_window")}), flush=True)
