#!/usr/bin/env python
"""Can a memory-bound kernel (LayerNorm, attention backward) run beside a weight-gradient GEMM that is held to part of the CUs?  The GEMM's
throughput barely depends on its block count (profiles/r4_wgrad_split_sweep.txt: the power envelope), so the CUs it leaves could carry the
HBM-bound kernels for free - if the envelope has room for both.  Two streams; alone / one after the other / side by side."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simseg_amd import ops  # noqa: E402
from simseg_amd.lib import call  # noqa: E402

MV = 512 * 197
g = torch.Generator(device="cuda").manual_seed(0)
dy = torch.randn(MV, 3072, device="cuda", generator=g).bfloat16()
x = torch.randn(MV, 768, device="cuda", generator=g).bfloat16()
dw = torch.zeros(3072, 768, device="cuda")
xf = torch.randn(MV, 768, device="cuda", generator=g)
gam, bet = torch.ones(768, device="cuda"), torch.zeros(768, device="cuda")
qkv = torch.randn(512, 197, 2304, device="cuda", generator=g).bfloat16()
o, lse = ops.attention_fwd(qkv, 12, None, save_lse=True)
do = torch.randn_like(o)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def wall(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


for blocks in (256, 192, 128, 96):
    sk = max(1, blocks // 36)
    call("simseg_debug_gemm_wgrad_blocks", blocks)

    def gemms(n=12):
        for _ in range(n):
            ops.gemm(dy, x, trans_a=True, trans_b=True, out=dw, accumulate=True, splitk=sk)

    for name, mem, n_mem in (("layernorm fwd x40", lambda: ops.layernorm_fwd(xf, gam, bet, 1e-6, out_dtype=torch.bfloat16), 40),
                             ("attention bwd x10", lambda: ops.attention_bwd(qkv, o, do, lse, 12, None), 10)):
        def mems():
            for _ in range(n_mem):
                mem()

        def both():
            with torch.cuda.stream(s1):
                gemms()
            with torch.cuda.stream(s2):
                mems()

        def g1():
            with torch.cuda.stream(s1):
                gemms()

        def m1():
            with torch.cuda.stream(s2):
                mems()
        for f in (g1, m1, both):
            f()
        tg, tm, tb = wall(g1), wall(m1), wall(both)
        print(f"wgrad 3072x768 x12 on {36 * sk:3d} blocks: {tg:6.2f} ms | {name}: {tm:6.2f} ms | side by side {tb:6.2f} ms (sum {tg + tm:6.2f}, saved {tg + tm - tb:5.2f})", flush=True)
