#!/usr/bin/env python
"""GEMM shapes with 100-300 tiles of 256x256 (the packed text tower: ~22 k rows x 768 columns): 128x128 kernel vs ping-pong kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simseg_amd import ops  # noqa: E402
from simseg_amd.graph import GraphedCall  # noqa: E402


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print(f"{'kind':<5}{'M':>7}{'N':>6}{'K':>6} {'tiles256':>9} {'128x128 us':>11} {'pingpong us':>12}")
for kind, M, N, K in [("nt", 21760, 768, 768), ("nt", 21760, 768, 3072), ("nn", 21760, 768, 768), ("nn", 21760, 768, 2304), ("nn", 21760, 768, 3072),
                      ("nt", 12288, 768, 768), ("nt", 12288, 768, 3072), ("nt", 12288, 2304, 768), ("nt", 8192, 3072, 768), ("nn", 8192, 768, 3072),
                      ("nt", 6144, 2304, 768), ("nt", 6144, 768, 3072)]:
    a = torch.randn(M, K, device="cuda").bfloat16()
    b = (torch.randn(K, N, device="cuda") if kind == "nn" else torch.randn(N, K, device="cuda")).bfloat16()
    res = {}
    for v in (1, 3):
        ops.set_gemm_variant(v)
        g = GraphedCall(lambda x: [ops.gemm(x, b, trans_b=(kind == "nn")) for _ in range(10)][-1], a)
        res[v] = timeit(lambda: g(a)) / 10
    ops.set_gemm_variant(0)
    print(f"{kind:<5}{M:>7}{N:>6}{K:>6} {((M + 255) // 256) * ((N + 255) // 256):>9} {res[1]:>11.1f} {res[3]:>12.1f}")
