#!/usr/bin/env python
"""Per-block timeline of the split-K weight-gradient GEMM (dW = dY^T . X, 256x256 ping-pong kernel, fp32 atomics): prologue / K loop /
epilogue (the atomic adds) of every block and the K loop's time per 64-deep K-tile.  argv: [M N K] (output M x N, contraction K)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simseg_amd import ops  # noqa: E402
from simseg_amd.lib import call, ptr  # noqa: E402
from simseg_amd.towers import _splitk  # noqa: E402

M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (2304, 768, 100864)
dy = torch.randn(K, M, device="cuda").bfloat16()
x = torch.randn(K, N, device="cuda").bfloat16()
out = torch.zeros(M, N, device="cuda")
sk = _splitk(M, N, K)
tiles = ((M + 255) // 256) * ((N + 255) // 256)
nk = K // 64
ks = (nk + sk - 1) // sk
z = (nk + ks - 1) // ks
blocks = tiles * z
buf = torch.zeros(blocks * 9 + 64, device="cuda", dtype=torch.int64)
kw = dict(trans_a=True, trans_b=True, out=out, accumulate=True, splitk=sk)
for _ in range(3):
    ops.gemm(dy, x, **kw)
call("simseg_debug_gemm_trace", ptr(buf))
ops.gemm(dy, x, **kw)
call("simseg_debug_gemm_trace", None)
torch.cuda.synchronize()
print("kernel:", ops.gemm_last_variant() if hasattr(ops, "gemm_last_variant") else "?")
raw = buf.cpu().numpy()
t = raw[:blocks * 5].reshape(blocks, 5)
ok = t[:, 0] > 0
t0 = t[ok, 0].min()
us = (t[:, :4] - t0) / 100.0
pro, loop, epi = us[:, 1] - us[:, 0], us[:, 2] - us[:, 1], us[:, 3] - us[:, 2]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.gemm(dy, x, **kw)
e1.record()
torch.cuda.synchronize()
print(f"dW {M}x{N}, K={K}: {tiles} tiles x {z} K-ranges of {ks} K-tiles = {blocks} blocks (traced {int(ok.sum())}); {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per launch untraced")
print(f"launch span {us[ok, 3].max():.1f} us; block start spread {us[ok, 0].max():.1f} us; per block: prologue {pro[ok].mean():.2f}, K loop {loop[ok].mean():.2f} "
      f"(p10 {np.percentile(loop[ok], 10):.1f}, p90 {np.percentile(loop[ok], 90):.1f}) = {loop[ok].mean() / ks:.3f} us per K-tile, epilogue (atomics) {epi[ok].mean():.2f} (p90 {np.percentile(epi[ok], 90):.2f}) us")
print(f"block end times: p10 {np.percentile(us[ok, 3], 10):.1f}  p50 {np.percentile(us[ok, 3], 50):.1f}  p90 {np.percentile(us[ok, 3], 90):.1f}  max {us[ok, 3].max():.1f} us")
