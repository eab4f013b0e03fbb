#!/usr/bin/env python
"""Batch-1 segmentation latency of one configuration (the call pattern of the reference's tool), for rocprofv3 runs.
argv: dtype(fp32|bf16) input classes encoder_tag dim [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

dtype, img, classes, tag, dim = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 20
print(bench.seg_latency_bench(torch.device("cuda", 0), dtype, img, classes, tag, dim, reps=reps))
