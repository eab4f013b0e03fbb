"""One seg-eval leg (ViT-S @512, bf16, 256 windows, no CRF: BASELINE configs[1]) for a rocprofv3 kernel trace:  rocprofv3 --kernel-trace -- python tools/seg_vits_prof.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
r = bench.seg_eval_bench(torch.device("cuda", 0), 1, "bf16", windows=256, img=512, classes=21, tag="vit_small_patch16_224_in21k", dim=384, steps=2)
print(json.dumps({k: r[k] for k in ("windows_per_s", "post_ms_per_step", "frac_of_peak")}))
