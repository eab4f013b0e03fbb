#!/usr/bin/env python
"""Round 6: the split-K weight-gradient GEMMs (dW = dY^T . X, 256x256 ping-pong kernel, fp32 atomics) in the dispatcher's
form and - SIMSEG_GEMM_WGRAD_XCD=1, read once per process - with every K-slice of an output tile on ONE XCD and L2-scope atomics.
Prints time, TFLOP/s and the error against fp32 torch per shape.  argv: [iters]   (rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE passes: tools/pmc_run.sh)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simseg_amd import ops  # noqa: E402
from simseg_amd.towers import _splitk  # noqa: E402

MV = 512 * 197
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
tag = "xcd-affine" if os.environ.get("SIMSEG_GEMM_WGRAD_XCD", "0") == "1" else "dispatcher"
for M, N, K in ((2304, 768, MV), (768, 768, MV), (3072, 768, MV), (768, 3072, MV)):
    torch.manual_seed(M + N)
    dy = torch.randn(K, M, device="cuda").bfloat16()
    x = torch.randn(K, N, device="cuda").bfloat16()
    out = torch.zeros(M, N, device="cuda")
    sk = _splitk(M, N, K)
    kw = dict(trans_a=True, trans_b=True, out=out, accumulate=True, splitk=sk)
    ops.gemm(dy, x, **kw)
    torch.cuda.synchronize()
    want = dy[:8192].float().t() @ x[:8192].float()
    chk = torch.zeros(M, N, device="cuda")
    ops.gemm(dy[:8192].contiguous(), x[:8192].contiguous(), trans_a=True, trans_b=True, out=chk, accumulate=True, splitk=max(2, min(sk, 8)))
    err = ((chk - want).abs().max() / want.abs().max()).item()
    for _ in range(3):
        ops.gemm(dy, x, **kw)
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            ops.gemm(dy, x, **kw)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    print(f"{tag:10s} dW {M:4d}x{N:4d} K={K} splitk {sk:2d}: {best:7.1f} us {2.0 * M * N * K / best / 1e6:6.0f} TFLOP/s  rel err (K=8192 check) {err:.2e}", flush=True)
