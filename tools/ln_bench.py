#!/usr/bin/env python
"""Micro-benchmark of the LayerNorm kernels at the training step's shape ([512*197, 768]); achieved GB/s of algorithmic traffic."""
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
    return e0.elapsed_time(e1) / iters * 1e-3


if __name__ == "__main__":
    M, D = 512 * 197, 768
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(M, D, device="cuda", generator=g)
    w, b = torch.ones(D, device="cuda"), torch.zeros(D, device="cuda")
    y, _, mean, rstd = ops.layernorm_fwd(x, w, b, 1e-6, out_dtype=torch.bfloat16, save_stats=True)
    s = timeit(lambda: ops.layernorm_fwd(x, w, b, 1e-6, out_dtype=torch.bfloat16, save_stats=True))
    print(f"ln_fwd  fp32 -> bf16   {s * 1e6:8.1f} us   {M * D * 6 / s / 1e9:8.1f} GB/s")
    dy16 = torch.randn(M, D, device="cuda", generator=g).bfloat16()
    dres = torch.randn(M, D, device="cuda", generator=g)
    dg, db, dsum = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    s = timeit(lambda: ops.layernorm_bwd(x, mean, rstd, w, dg, db, dy16=dy16, dres=dres, dxsum=dsum))
    print(f"ln_bwd  (x, dy16, dres) -> (dx32, dx16)   {s * 1e6:8.1f} us   {M * D * 16 / s / 1e9:8.1f} GB/s")
