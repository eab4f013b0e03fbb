#!/usr/bin/env python
"""Per-block timeline of the 256x256 ping-pong GEMM: where a tile's time goes (prologue / K loop / epilogue) and how the blocks of a
launch line up in time.  argv: [M N K] [nt|nn]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simseg_amd import ops  # noqa: E402
from simseg_amd.lib import call, ptr  # noqa: E402

M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (100864, 2304, 768)
kind = sys.argv[4] if len(sys.argv) > 4 else "nt"
stagger = int(sys.argv[5]) if len(sys.argv) > 5 else 0
act = int(sys.argv[6]) if len(sys.argv) > 6 else 0
kw = {}
if act == 3:
    kw = dict(act=3, bias=torch.zeros(N, device="cuda"), aux_out=torch.empty(M, N, device="cuda", dtype=torch.bfloat16))
elif act == 4:
    kw = dict(act=4, aux=torch.randn(M, N, device="cuda").bfloat16(), colsum=torch.zeros(N, device="cuda"))
elif act == 13:      # simseg_gemm act 5: GELU stored, GELU' saved as the tile-blocked accumulator image
    kw = dict(act=5, bias=torch.zeros(N, device="cuda"), aux_out=torch.empty(M, N, device="cuda", dtype=torch.bfloat16))
elif act == 14:      # simseg_gemm act 6: times the tile-blocked saved derivative + column sums
    kw = dict(act=6, aux=torch.randn(M, N, device="cuda").bfloat16(), colsum=torch.zeros(N, device="cuda"))
elif act == 5:
    kw = dict(bias=torch.zeros(N, device="cuda"), residual=torch.randn(M, N, device="cuda"), out_dtype=torch.float32)
call("simseg_debug_gemm_stagger", stagger)
a = torch.randn(M, K, device="cuda").bfloat16()
b = (torch.randn(K, N, device="cuda") if kind == "nn" else torch.randn(N, K, device="cuda")).bfloat16()
tiles = ((M + 255) // 256) * ((N + 255) // 256)
buf = torch.zeros(tiles * 9, device="cuda", dtype=torch.int64)
ops.set_gemm_variant(3)
for _ in range(3):
    ops.gemm(a, b, trans_b=(kind == "nn"), **kw)
call("simseg_debug_gemm_trace", ptr(buf))
ops.gemm(a, b, trans_b=(kind == "nn"), **kw)
call("simseg_debug_gemm_trace", None)
torch.cuda.synchronize()
raw = buf.cpu().numpy()
t = raw[:tiles * 5].reshape(tiles, 5)
sub = raw[tiles * 5:].reshape(tiles, 4)
t0 = t[:, 0].min()
us = (t[:, :4] - t0) / 100.0
issued = (t[:, 4] - t0) / 100.0
pro, loop, epi = us[:, 1] - us[:, 0], us[:, 2] - us[:, 1], us[:, 3] - us[:, 2]
subus = (sub - t0) / 100.0
print("epilogue of wave 0, start of each of its four 32x64 blocks after the K loop's end:", np.round((subus - us[:, 2:3]).mean(0), 2), "us; end", round(float((issued - us[:, 2]).mean()), 2))
print(f"epilogue split (wave 0): until its last store is issued {(issued - us[:, 2]).mean():.2f} us, then until acknowledged {(us[:, 3] - issued).mean():.2f} us")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.gemm(a, b, trans_b=(kind == "nn"), **kw)
e1.record()
torch.cuda.synchronize()
print(f"stagger {stagger} ticks: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per launch (untraced)")
print(f"{M}x{N}x{K} {kind}: {tiles} tiles, launch span {us[:, 3].max():.1f} us; per tile: prologue {pro.mean():.2f} (p90 {np.percentile(pro, 90):.2f}), "
      f"K loop {loop.mean():.2f} (p90 {np.percentile(loop, 90):.2f}), epilogue {epi.mean():.2f} (p90 {np.percentile(epi, 90):.2f}), total {(us[:, 3] - us[:, 0]).mean():.2f} us")
# how synchronised are the blocks: fraction of blocks that are in their epilogue at each instant
grid = np.arange(0, us[:, 3].max(), 1.0)
in_epi = np.array([((us[:, 2] <= g) & (us[:, 3] > g)).sum() for g in grid])
in_loop = np.array([((us[:, 1] <= g) & (us[:, 2] > g)).sum() for g in grid])
active = np.array([((us[:, 0] <= g) & (us[:, 3] > g)).sum() for g in grid])
print("time(us): active / in K loop / in epilogue  (every 4 us)")
for i in range(0, len(grid), 4):
    print(f"  {grid[i]:7.0f}: {active[i]:4d} {in_loop[i]:4d} {in_epi[i]:4d}")
if stagger:
    for grp in range(4):
        sel = [b for b in range(min(256, tiles)) if ((b >> 3) & 3) == grp]
        print(f"first round, stagger group {grp}: start {us[sel, 0].mean():.1f} us, K loop {loop[sel].mean():.2f}, epilogue {epi[sel].mean():.2f} us")
