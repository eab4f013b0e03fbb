#!/usr/bin/env python
"""Micro-benchmark of simseg_gemm on the shapes of the ViT-B / BERT-base training step (512 pairs per GPU).
    python tools/gemm_bench.py [--shapes train|quick] [--iters 20] [--only nt|nn|tn]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simseg_amd import ops  # noqa: E402

MV, MB = 512 * 197, 512 * 77
SHAPES = {
    "train": [
        ("nt", MV, 2304, 768), ("nt", MV, 768, 768), ("nt", MV, 3072, 768), ("nt", MV, 768, 3072), ("nt", MB, 2304, 768), ("nt", MB, 3072, 768),
        ("nn", MV, 768, 2304), ("nn", MV, 768, 768), ("nn", MV, 3072, 768), ("nn", MV, 768, 3072), ("nn", MB, 768, 3072),
        ("tn", 2304, 768, MV), ("tn", 768, 768, MV), ("tn", 3072, 768, MV), ("tn", 768, 3072, MV), ("tn", 3072, 768, MB),
    ],
    "quick": [("nt", MV, 3072, 768), ("nt", MV, 768, 3072), ("nn", MV, 768, 3072), ("tn", 3072, 768, MV)],
    "nsplit": [("nt", MV, 2304, 768), ("nt", MV, 1152, 768), ("nt", MV, 3072, 768), ("nt", MV, 1536, 768), ("nt", MV, 1024, 768),
               ("nn", MV, 3072, 768), ("nn", MV, 1536, 768), ("nt", MV, 768, 768), ("nt", MV, 512, 768), ("nt", MV, 256, 768)],
    "square": [("nt", 4096, 4096, 4096), ("nt", 8192, 8192, 8192)],
    "msweep": [("nt", m, 768, 128) for m in (256, 4096, 16384, 65536, MV)] + [("nt", m, 768, 768) for m in (256, 4096, 16384, 65536, MV)],
    "ksweep": [("nt", MV, 768, k) for k in (128, 256, 512, 768, 1536, 3072, 6144)],
}


def run_vendor(kind, M, N, K, iters):
    """Yardstick only (never used by the product path): the vendor BLAS through torch.matmul on the same operands/layouts."""
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    ta, tb = kind == "tn", kind in ("nn", "tn")
    a = torch.randn((K, M) if ta else (M, K), device=dev, generator=g).bfloat16()
    b = torch.randn((K, N) if tb else (N, K), device=dev, generator=g).bfloat16()
    A = a.t() if ta else a
    Bm = b if tb else b.t()
    for _ in range(3):
        torch.matmul(A, Bm)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        torch.matmul(A, Bm)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, 2.0 * M * N * K / ms / 1e9


def run(kind, M, N, K, iters, out_f32=False, act=0, colsum=False):
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    ta, tb = kind == "tn", kind in ("nn", "tn")
    a = torch.randn((K, M) if ta else (M, K), device=dev, generator=g).bfloat16()
    b = torch.randn((K, N) if tb else (N, K), device=dev, generator=g).bfloat16()
    kw = dict(trans_a=ta, trans_b=tb)
    if kind == "tn":
        out = torch.zeros(M, N, device=dev)
        tiles = ((M + 127) // 128) * ((N + 127) // 128)
        kw.update(out=out, accumulate=True, splitk=max(1, min((K + 63) // 64, (1024 + tiles - 1) // tiles, 64)))
    elif out_f32:
        kw.update(out_dtype=torch.float32)
    if kind != "tn":
        if act == 1:        # forward fc1: bias + GELU, pre-activation saved
            kw.update(act=1, bias=torch.zeros(N, device=dev), aux_out=torch.empty(M, N, device=dev, dtype=torch.bfloat16))
        elif act == 2:      # dgrad through fc2: multiply by GELU'(pre-activation)
            kw.update(act=2, aux=torch.randn(M, N, device=dev, generator=g).bfloat16())
        elif act == 3:      # forward fc1 as the training step runs it: bias + GELU, GELU' saved
            kw.update(act=3, bias=torch.zeros(N, device=dev), aux_out=torch.empty(M, N, device=dev, dtype=torch.bfloat16))
        elif act == 4:      # dgrad through fc2 as the training step runs it: times the saved derivative
            kw.update(act=4, aux=torch.randn(M, N, device=dev, generator=g).bfloat16())
        elif act == 7:      # fc1 forward with GELU' saved as the tile-blocked accumulator image (what the step runs on full tiles)
            kw.update(act=5, bias=torch.zeros(N, device=dev), aux_out=torch.empty(M, N, device=dev, dtype=torch.bfloat16))
        elif act == 8:      # dgrad through fc2 reading that image
            kw.update(act=6, aux=torch.randn(M, N, device=dev, generator=g).bfloat16())
        elif act == 5:      # proj / fc2 forward: bias + fp32 residual, fp32 output
            kw.update(bias=torch.zeros(N, device=dev), residual=torch.randn(M, N, device=dev, generator=g), out_dtype=torch.float32)
        if colsum:
            kw.update(colsum=torch.zeros(N, device=dev))
    for _ in range(3):
        ops.gemm(a, b, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.gemm(a, b, **kw)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, 2.0 * M * N * K / ms / 1e9


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="train")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default=None)
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--act", type=int, default=0, help="1: bias+GELU with pre-activation save; 2: times GELU'(aux)")
    ap.add_argument("--colsum", action="store_true", help="also accumulate column sums of the output (fused bias gradient)")
    ap.add_argument("--vendor", action="store_true", help="time torch.matmul (vendor BLAS) instead, as a yardstick")
    args = ap.parse_args()
    ops.set_gemm_variant(args.variant)
    tot_ms = tot_fl = 0.0
    for kind, M, N, K in SHAPES[args.shapes]:
        if args.only and kind != args.only:
            continue
        ms, tf = run_vendor(kind, M, N, K, args.iters) if args.vendor else run(kind, M, N, K, args.iters, act=args.act, colsum=args.colsum)
        tot_ms += ms; tot_fl += 2.0 * M * N * K
        print(f"{kind} M={M:6d} N={N:5d} K={K:6d}  {ms:8.3f} ms  {tf:8.1f} TFLOP/s", flush=True)
    print(f"sum {tot_ms:.3f} ms  -> {tot_fl / tot_ms / 1e9:.1f} TFLOP/s aggregate")
