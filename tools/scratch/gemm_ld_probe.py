#!/usr/bin/env python
"""Does the leading dimension of the K-contiguous operands (x . W^T: both [rows, K] row-major) matter?  A 256-row operand tile is 256
pieces of 128 B one leading dimension apart: with K = 768 (1536 B) or a power of two they may crowd a few L2 / memory channels.  The
same NT launches with lda / ldb padded by 64 / 72 elements.  Raw simseg_gemm calls (ops.gemm takes contiguous operands only)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from simseg_amd import ops  # noqa: E402
from simseg_amd.lib import call, ptr  # noqa: E402
from simseg_amd.ops import stream  # noqa: E402


def run(M, N, K, lda, ldb, zeros=False, iters=20):
    g = torch.Generator(device="cuda").manual_seed(0)
    mk = (lambda *s: torch.zeros(*s, device="cuda")) if zeros else (lambda *s: torch.randn(*s, device="cuda", generator=g))
    a = mk(M, lda).bfloat16()
    b = mk(N, ldb).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)

    def go():
        call("simseg_gemm", ptr(a), ptr(b), ptr(out), M, N, K, lda, ldb, N, 1, 1, 0, 0, 1.0, None, None, None, 0, 0, None, None, 0, 0, 0, 1, 0, 0.0, None, stream())
    for _ in range(5):
        go()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        go()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms * 1e3, 2.0 * M * N * K / ms / 1e9


for M, N, K in ((100864, 3072, 768), (100864, 2304, 768), (100864, 768, 768), (100864, 768, 3072), (8192, 8192, 8192)):
    for zeros in (False, True):
        row = []
        for pa, pb in ((0, 0), (0, 64), (64, 0), (64, 64), (72, 72), (0, 8)):
            us, tf = run(M, N, K, K + pa, K + pb, zeros)
            row.append(f"lda+{pa:<2d} ldb+{pb:<2d}: {us:7.1f} us {tf:5.0f} TF")
        print(f"nt {M}x{N}x{K} {'zeros   ' if zeros else 'gaussian'} | " + " | ".join(row), flush=True)
