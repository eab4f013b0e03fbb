#!/usr/bin/env python
"""Split-K sweep of the weight-gradient GEMMs of the training step (dW = dY^T . X on the 256x256 ping-pong kernel, fp32 atomics):
launch time against the number of K-ranges, i.e. blocks = tiles x ranges against the 256 CUs.  argv: [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simseg_amd import ops  # noqa: E402
from simseg_amd.lib import call  # noqa: E402
from simseg_amd.towers import _splitk  # noqa: E402

MV, MB, MT = 512 * 197, 512 * 77, 22016
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for M, N, K in ((2304, 768, MV), (768, 768, MV), (3072, 768, MV), (768, 3072, MV), (2304, 768, MT), (3072, 768, MT), (768, 3072, MT), (768, 768, MT), (512, 768, MV)):
    dy = torch.randn(K, M, device="cuda").bfloat16()
    x = torch.randn(K, N, device="cuda").bfloat16()
    out = torch.zeros(M, N, device="cuda")
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    nk = K // 64
    cur = _splitk(M, N, K)
    cands = sorted({cur, max(1, 256 // tiles), max(1, 256 // tiles - 1), max(1, 256 // tiles + 1), max(1, 512 // tiles), max(1, 128 // tiles)})
    row = []
    for sk in cands:
        call("simseg_debug_gemm_wgrad_blocks", tiles * sk)      # (the dispatcher otherwise picks one round of all CUs whatever the caller asks for)
        kw = dict(trans_a=True, trans_b=True, out=out, accumulate=True, splitk=sk)
        for _ in range(3):
            ops.gemm(dy, x, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            ops.gemm(dy, x, **kw)
        e1.record()
        torch.cuda.synchronize()
        ks = (nk + sk - 1) // sk
        z = (nk + ks - 1) // ks
        us = e0.elapsed_time(e1) / iters * 1e3
        row.append(f"{'*' if sk == cur else ' '}splitk {sk:3d} -> {tiles * z:4d} blocks {us:7.1f} us {2.0 * M * N * K / us / 1e6:6.0f} TF")
    print(f"dW {M}x{N} K={K} ({tiles} tiles): " + " | ".join(row), flush=True)
