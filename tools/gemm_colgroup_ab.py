#!/usr/bin/env python
"""Tile order of the persistent 256x256 GEMM (simseg_debug_gemm_colgroup): row-major over all column tiles (0) against column groups (auto
and forced widths) on the wide-N shapes of the training step, interleaved in one process.
    python tools/gemm_colgroup_ab.py          rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python tools/gemm_colgroup_ab.py pmc   (one launch set per order)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simseg_amd import ops  # noqa: E402
from simseg_amd.lib import call  # noqa: E402

MV = 512 * 197
SHAPES = [("nt", MV, 3072, 768, 7), ("nt", MV, 2304, 768, 0), ("nn", MV, 3072, 768, 0), ("nn", MV, 3072, 768, 8), ("nn", MV, 768, 3072, 0), ("nt", MV, 768, 3072, 0), ("nt", 512 * 64, 3072, 768, 0)]


def make(kind, M, N, K, act):
    g = torch.Generator(device="cuda").manual_seed(0)
    a = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    b = (torch.randn(K, N, device="cuda", generator=g) * 0.02).bfloat16() if kind == "nn" else (torch.randn(N, K, device="cuda", generator=g) * 0.02).bfloat16()
    kw = dict(trans_b=(kind == "nn"), out_dtype=torch.bfloat16)
    if act == 7:
        kw.update(act=7, bias=torch.zeros(N, device="cuda"), aux_out=torch.empty(M * N, device="cuda", dtype=torch.uint8))
    if act == 8:
        kw.update(act=8, aux=torch.randint(0, 255, (M * N,), device="cuda", dtype=torch.uint8))
    return lambda: ops.gemm(a, b, **kw)


def timeit(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


if __name__ == "__main__":
    pmc = len(sys.argv) > 1 and sys.argv[1] == "pmc"
    for kind, M, N, K, act in SHAPES:
        fn = make(kind, M, N, K, act)
        orders = (0, -1) if pmc else (0, -1, 3, 4, 6)
        res = {o: [] for o in orders}
        for rep in range(1 if pmc else 3):
            for o in orders:
                call("simseg_debug_gemm_colgroup", o)
                fn(); fn()
                res[o].append(timeit(fn, 4 if pmc else 20))
        call("simseg_debug_gemm_colgroup", -1)
        fl = 2.0 * M * N * K
        print(f"{kind} M={M} N={N} K={K} act={act}: " + " | ".join(f"colgroup {o:2d}: {min(v):7.1f} us {fl / min(v) / 1e6:6.0f} TF" for o, v in res.items()), flush=True)
