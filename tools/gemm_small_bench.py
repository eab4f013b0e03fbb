#!/usr/bin/env python
"""Small-problem GEMM shapes (batch-1 segmentation: 325 / 1025 token rows) on the 128x128 kernel vs the small-problem kernel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from simseg_amd import ops  # noqa: E402
from simseg_amd.graph import GraphedCall  # noqa: E402

SHAPES = [("fp32", 325, 384, 384), ("fp32", 325, 1152, 384), ("fp32", 325, 1536, 384), ("fp32", 325, 384, 1536),
          ("fp32", 1025, 768, 768), ("fp32", 1025, 2304, 768), ("fp32", 1025, 3072, 768), ("fp32", 1025, 768, 3072),
          ("bf16", 1025, 768, 768), ("bf16", 1025, 2304, 768), ("bf16", 1025, 3072, 768), ("bf16", 1025, 768, 3072),
          ("bf16", 325, 384, 384), ("bf16", 325, 1536, 384), ("bf16", 2050, 768, 768), ("bf16", 4100, 768, 768), ("bf16", 4100, 3072, 768)]


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print(f"{'dtype':<6}{'M':>6}{'N':>6}{'K':>6} {'128x128 us':>12} {'small us':>10} {'ring128 us':>10} {'speedup':>8} {'TFLOP/s':>8}")
for dt, M, N, K in SHAPES:
    t = torch.float32 if dt == "fp32" else torch.bfloat16
    a = torch.randn(M, K, device="cuda").to(t)
    b = torch.randn(N, K, device="cuda").to(t)
    bias = torch.randn(N, device="cuda")
    res = {}
    for v in (1, 4, 5):
        if v == 5 and dt == "fp32":
            res[v] = float("nan")
            continue
        ops.set_gemm_variant(v)
        g = GraphedCall(lambda x: [ops.gemm(x, b, bias=bias) for _ in range(20)][-1], a)       # 20 launches per replay: GPU time, not Python's
        res[v] = timeit(lambda: g(a), reps=10) / 20
    ops.set_gemm_variant(0)
    print(f"{dt:<6}{M:>6}{N:>6}{K:>6} {res[1]:>12.1f} {res[4]:>10.1f} {res[5]:>10.1f} {res[1] / res[4]:>8.2f} {2.0 * M * N * K / res[4] / 1e6:>8.1f}")
