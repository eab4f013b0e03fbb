#!/usr/bin/env python
"""Debug: a few launches of ONE attention configuration (for rocprofv3 --pmc runs).  argv: T variant [bwd]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simseg_amd import ops  # noqa: E402

T, variant = int(sys.argv[1]), int(sys.argv[2])
bwd = len(sys.argv) > 3
B, H = (512, 12) if T <= 256 else (16, 12)
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B, T, 3 * H * 64, device="cuda", generator=g).bfloat16()
ops.set_attention_variant(variant)
out, lse = ops.attention_fwd(qkv, H, None, save_lse=True)
for _ in range(3):
    if bwd:
        ops.attention_bwd(qkv, out, torch.randn_like(out), lse, H, None)
    else:
        ops.attention_fwd(qkv, H, None, save_lse=True)
torch.cuda.synchronize()
