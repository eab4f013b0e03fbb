#!/usr/bin/env python
"""Within-process A/B of simseg_gemm kernel variants (cdna_hip_programming.md 5.4 rule 24: N variants x M rounds interleaved in ONE
process, median and min reported).
    python tools/gemm_ab.py --variants 3,10,11,12,13 [--shapes train] [--rounds 7] [--iters 5] [--only nt,nn] [--act 0]"""
import argparse
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simseg_amd import ops  # noqa: E402
from tools.gemm_bench import SHAPES  # noqa: E402

SHAPES["kloop"] = [("nt", 100864, 768, 3072), ("nn", 100864, 768, 3072), ("nt", 8192, 8192, 8192), ("nt", 100864, 2304, 768), ("nn", 100864, 3072, 768)]


def make(kind, M, N, K, act):
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    tb = kind == "nn"
    a = torch.randn(M, K, device=dev, generator=g).bfloat16()
    b = torch.randn((K, N) if tb else (N, K), device=dev, generator=g).bfloat16()
    kw = dict(trans_b=tb)
    if act == 3:
        kw.update(act=3, bias=torch.zeros(N, device=dev), aux_out=torch.empty(M, N, device=dev, dtype=torch.bfloat16))
    elif act == 4:
        kw.update(act=4, aux=torch.randn(M, N, device=dev, generator=g).bfloat16(), colsum=torch.zeros(N, device=dev))
    elif act == 5:
        kw.update(bias=torch.zeros(N, device=dev), residual=torch.randn(M, N, device=dev, generator=g), out_dtype=torch.float32)
    kw["out"] = torch.empty(M, N, device=dev, dtype=kw.pop("out_dtype", torch.bfloat16))
    return a, b, kw


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="15,0")
    ap.add_argument("--shapes", default="train")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--only", default="nt,nn")
    ap.add_argument("--act", type=int, default=0)
    args = ap.parse_args()
    variants = [int(v) for v in args.variants.split(",")]
    kinds = args.only.split(",")
    tot = {v: 0.0 for v in variants}
    flops = 0.0
    print(f"# act={args.act} rounds={args.rounds} iters={args.iters}; median ms (min ms) TFLOP/s per variant")
    for kind, M, N, K in SHAPES[args.shapes]:
        if kind not in kinds:
            continue
        a, b, kw = make(kind, M, N, K, args.act)
        times = {v: [] for v in variants}
        for v in variants:
            ops.set_gemm_variant(v)
            ops.gemm(a, b, **kw)
        for r in range(args.rounds):
            for v in (variants if r % 2 == 0 else variants[::-1]):
                ops.set_gemm_variant(v)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    ops.gemm(a, b, **kw)
                e1.record()
                torch.cuda.synchronize()
                times[v].append(e0.elapsed_time(e1) / args.iters)
        fl = 2.0 * M * N * K
        flops += fl
        line = f"{kind} M={M:6d} N={N:5d} K={K:5d} "
        for v in variants:
            med, mn = statistics.median(times[v]), min(times[v])
            tot[v] += med
            line += f" | v{v}: {med:.3f} ({mn:.3f}) {fl / med / 1e9:6.0f}"
        print(line, flush=True)
        del a, b, kw
    ops.set_gemm_variant(0)
    print("sum of medians:" + "".join(f" | v{v}: {tot[v]:.3f} ms {flops / tot[v] / 1e9:6.0f} TF" for v in variants))
