#!/usr/bin/env python
"""Debug: where the resident attention forward spends its time (variants: full, no K/V copy, no tile loop, ring kernel)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simseg_amd import ops  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


g = torch.Generator(device="cuda").manual_seed(0)
from simseg_amd.lib import raw
print("predicted resident blocks per CU:", {T: raw("simseg_debug_attn_occupancy", T) for T in (25, 77, 197, 256)})
for B, T, H in ((512, 197, 12), (512, 77, 12), (512, 25, 12)):
    qkv = torch.randn(B, T, 3 * H * 64, device="cuda", generator=g).bfloat16()
    row = [f"B={B} T={T}"]
    for v, name in ((4, "resident"), (102, "no-copy"), (103, "no-loop"), (1, "ring")):
        ops.set_attention_variant(v)
        row.append(f"{name} {timeit(lambda: ops.attention_fwd(qkv, H, None, save_lse=True)) * 1e3:.1f} us")
    ops.set_attention_variant(0)
    print("  ".join(row), flush=True)
