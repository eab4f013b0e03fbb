#!/usr/bin/env python
"""The forward GEMMs y = x . W^T of the training step as they run now (NT: W stored [out, in], both operands K-contiguous) against the
same products on a TRANSPOSED 16-bit weight copy (NN: W^T stored [in, out], the kernel the dgrads use).  Same box, interleaved rounds,
median us.  Epilogues: plain bf16 store (qkv), fp32 residual (proj / fc2)."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simseg_amd import ops  # noqa: E402

MV, MT = 512 * 197, 22016
g = torch.Generator(device="cuda").manual_seed(0)
tot = {"nt": 0.0, "nn": 0.0}
for name, M, N, K, f32res in (("vit qkv", MV, 2304, 768, False), ("vit proj", MV, 768, 768, True), ("vit fc1 (plain store)", MV, 3072, 768, False),
                              ("vit fc2", MV, 768, 3072, True), ("bert qkv", MT, 2304, 768, False), ("bert fc1 (plain store)", MT, 3072, 768, False),
                              ("bert fc2", MT, 768, 3072, True), ("bert out", MT, 768, 768, True)):
    a = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.02).bfloat16()
    wt = w.t().contiguous()
    kw = {}
    if f32res:
        kw = dict(bias=torch.zeros(N, device="cuda"), residual=torch.randn(M, N, device="cuda", generator=g), out=torch.empty(M, N, device="cuda"))
    else:
        kw = dict(bias=torch.zeros(N, device="cuda"), out=torch.empty(M, N, device="cuda", dtype=torch.bfloat16))
    fns = {"nt": lambda: ops.gemm(a, w, **kw), "nn": lambda: ops.gemm(a, wt, trans_b=True, **kw)}
    o1 = fns["nt"]().float().clone()
    o2 = fns["nn"]().float()
    same = torch.equal(o1, o2)
    times = {"nt": [], "nn": []}
    for r in range(6):
        for k in (("nt", "nn") if r % 2 == 0 else ("nn", "nt")):
            for _ in range(3):
                fns[k]()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fns[k]()
            e1.record()
            torch.cuda.synchronize()
            times[k].append(e0.elapsed_time(e1) * 100)
    nt, nn = statistics.median(times["nt"]), statistics.median(times["nn"])
    tot["nt"] += nt; tot["nn"] += nn
    fl = 2.0 * M * N * K
    print(f"{name:<24} M={M:6d} N={N:4d} K={K:4d}: NT {nt:7.1f} us {fl / nt / 1e6:5.0f} TF | NN {nn:7.1f} us {fl / nn / 1e6:5.0f} TF | {100 * (nn / nt - 1):+5.1f} %  bit-equal {same}", flush=True)
print(f"sum: NT {tot['nt']:.1f} us, NN {tot['nn']:.1f} us ({100 * (tot['nn'] / tot['nt'] - 1):+.1f} %)")
