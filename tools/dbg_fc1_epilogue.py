#!/usr/bin/env python
"""Where the fc1-forward epilogue's time goes (act 5: GELU stored, GELU' saved tile-blocked), by ablation on the persistent kernel:
plain store / no derivative store / no GELU arithmetic / no output store.  WRONG results in the ablated runs - timing only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simseg_amd import ops  # noqa: E402
from simseg_amd.lib import call  # noqa: E402

M, N, K = 512 * 197, 3072, 768
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.randn(M, K, device="cuda", generator=g).bfloat16()
b = torch.randn(N, K, device="cuda", generator=g).bfloat16()
bias = torch.zeros(N, device="cuda")
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
aux = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)


def t(fn, iters=20):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for rnd in range(3):
    row = []
    call("simseg_debug_gemm_stagger", 0)
    row.append(("act 0 plain", t(lambda: ops.gemm(a, b, out=out, bias=bias))))
    row.append(("act 1 gelu + pre saved row-major", t(lambda: ops.gemm(a, b, out=out, bias=bias, act=1, aux_out=aux))))
    row.append(("act 3 row-major", t(lambda: ops.gemm(a, b, out=out, bias=bias, act=3, aux_out=aux))))
    row.append(("act 5 blocked", t(lambda: ops.gemm(a, b, out=out, bias=bias, act=5, aux_out=aux))))
    for code, label in ((1001, "act 5 - no derivative store"), (1002, "act 5 - no GELU arithmetic"), (1003, "act 5 - no output store")):
        call("simseg_debug_gemm_stagger", code)
        row.append((label, t(lambda: ops.gemm(a, b, out=out, bias=bias, act=5, aux_out=aux))))
    call("simseg_debug_gemm_stagger", 0)
    print(" | ".join(f"{k}: {v:.3f}" for k, v in row), flush=True)
