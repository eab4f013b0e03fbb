#!/usr/bin/env python
"""Micro-benchmark of the LayerNorm kernels at the training step's shapes (ViT-B [512*197, 768]; ViT-S [512*197, 384]); achieved GB/s of
algorithmic traffic per form.  Forms of the backward:
    r2      (x fp32, dy16, dres fp32) -> (dx32, dx16)                    16 B / element   (round 2's form)
    dres16  (x fp32, dy16, dres16)    -> dx16                            10 B / element   (round 4: 16-bit residual gradient)
    y16     (y16, dy16, dres16)       -> dx16                             8 B / element   (round 4: xhat from the saved 16-bit output; the step's form)
    y16mix  as y16 with 1 % of the gains below the y16 test's threshold: those chunks read x
A/B of two builds: run once per library (SIMSEG_AMD_LIB=simseg_amd/libsimseg_hip_r4.so python tools/ln_bench.py)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simseg_amd import lib, ops  # noqa: E402


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def bench(M, D):
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(M, D, device="cuda", generator=g)
    w, b = torch.ones(D, device="cuda"), torch.zeros(D, device="cuda")
    y, _, mean, rstd = ops.layernorm_fwd(x, w, b, 1e-6, out_dtype=torch.bfloat16, save_stats=True)
    s = timeit(lambda: ops.layernorm_fwd(x, w, b, 1e-6, out_dtype=torch.bfloat16, save_stats=True))
    print(f"[{M},{D}] ln_fwd  fp32 -> bf16                           {s * 1e6:8.1f} us   {M * D * 6 / s / 1e9:8.1f} GB/s")
    dy16 = torch.randn(M, D, device="cuda", generator=g).bfloat16()
    dres = torch.randn(M, D, device="cuda", generator=g)
    dres16 = dres.bfloat16()
    dg, db, dsum = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    s = timeit(lambda: ops.layernorm_bwd(x, mean, rstd, w, dg, db, dy16=dy16, dres=dres, dxsum=dsum))
    print(f"[{M},{D}] ln_bwd  r2     (x, dy16, dres) -> (dx32, dx16)  {s * 1e6:8.1f} us   {M * D * 16 / s / 1e9:8.1f} GB/s")
    s = timeit(lambda: ops.layernorm_bwd(x, mean, rstd, w, dg, db, dy16=dy16, dres16=dres16, dxsum=dsum, want_f32=False))
    print(f"[{M},{D}] ln_bwd  dres16 (x, dy16, dres16) -> dx16        {s * 1e6:8.1f} us   {M * D * 10 / s / 1e9:8.1f} GB/s")
    s = timeit(lambda: ops.layernorm_bwd(x, mean, rstd, w, dg, db, dy16=dy16, dres16=dres16, dxsum=dsum, want_f32=False, y16=y, beta=b))
    print(f"[{M},{D}] ln_bwd  y16    (y16, dy16, dres16) -> dx16      {s * 1e6:8.1f} us   {M * D * 8 / s / 1e9:8.1f} GB/s")
    w2 = w.clone()
    w2[torch.randperm(D, device="cuda", generator=g)[: max(1, D // 100)]] = 0.01
    y2, _, mean, rstd = ops.layernorm_fwd(x, w2, b, 1e-6, out_dtype=torch.bfloat16, save_stats=True)
    s = timeit(lambda: ops.layernorm_bwd(x, mean, rstd, w2, dg, db, dy16=dy16, dres16=dres16, dxsum=dsum, want_f32=False, y16=y2, beta=b))
    print(f"[{M},{D}] ln_bwd  y16mix (1 % of the gains small)          {s * 1e6:8.1f} us   {M * D * 8 / s / 1e9:8.1f} GB/s (of the 8 B form)")


if __name__ == "__main__":
    print("library:", os.path.basename(lib.LIB_PATH), " SIMSEG_LN_BWD_BLOCKS_PER_CU =", os.environ.get("SIMSEG_LN_BWD_BLOCKS_PER_CU", "(occupancy)"))
    bench(512 * 197, 768)
    bench(512 * 197, 384)
