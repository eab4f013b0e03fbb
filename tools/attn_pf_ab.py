#!/usr/bin/env python
"""A/B of the one-kernel attention backward with (variant 0) / without (variant 4) the next-head operand touches, interleaved in one process.
    python tools/attn_pf_ab.py [B T H]"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simseg_amd import ops  # noqa: E402

cases = [(512, 197, 12, False), (512, 77, 12, True)] if len(sys.argv) < 4 else [(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), False)]
for B, T, H, masked in cases:
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = (torch.randn(B, T, 3 * H * 64, device="cuda", generator=g) * 0.8).bfloat16()
    dout = torch.randn(B, T, H * 64, device="cuda", generator=g).bfloat16()
    mask = None
    if masked:
        lens = torch.randint(8, T + 1, (B,), device="cuda", generator=g)
        mask = (torch.arange(T, device="cuda")[None] < lens[:, None]).long()
    out, lse = ops.attention_fwd(qkv, H, mask, save_lse=True)
    res = {}
    VAR = (0, 4, 5, 6, 7)
    times = {v: [] for v in VAR}
    for v in VAR:
        ops.set_attention_variant(v)
        res[v] = ops.attention_bwd(qkv, out, dout, lse, H, mask)
    assert all(torch.equal(res[0], res[v]) for v in VAR)
    for r in range(9):
        for v in (VAR if r % 2 == 0 else VAR[::-1]):
            ops.set_attention_variant(v)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.attention_bwd(qkv, out, dout, lse, H, mask)
            e1.record()
            torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / 10)
    ops.set_attention_variant(0)
    print(f"B={B} T={T} H={H} masked={masked}: " + "  ".join(f"v{v}: {statistics.median(times[v]):.4f} ({min(times[v]):.4f})" for v in VAR) +
          "   [0 = all five operands touched, 4 = none, 5 = q k v, 6 = q k, 7 = q k v dO]", flush=True)
