#!/usr/bin/env python
"""The forward GEMM shapes of ViT-S (D = 384) at the seg-eval batch (256 windows of 512^2: M = 262400 rows) on every kernel variant that
accepts them: which kernel the dispatcher picks (variant 0) against the forced ones.   python tools/gemm_vits_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simseg_amd import ops  # noqa: E402
from simseg_amd.lib import raw  # noqa: E402


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


if __name__ == "__main__":
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 262400
    g = torch.Generator(device="cuda").manual_seed(0)
    for name, N, K, res in (("qkv", 1152, 384, False), ("proj", 384, 384, True), ("fc1", 1536, 384, False), ("fc2", 384, 1536, True)):
        a = torch.randn(M, K, device="cuda", generator=g).bfloat16()
        w = (torch.randn(N, K, device="cuda", generator=g) * 0.02).bfloat16()
        bias = torch.zeros(N, device="cuda")
        kw = dict(bias=bias)
        if res:
            kw.update(residual=torch.randn(M, N, device="cuda", generator=g), out_dtype=torch.float32)
        elif name == "fc1":
            kw.update(act=1)            # GELU (evaluation forward: nothing saved)
        by = M * K * 2 + N * K * 2 + M * N * (8 if res else 2)
        line = f"{name:5s} M={M} N={N} K={K} {'fp32 out + residual' if res else 'bf16 out':20s}"
        ops.set_gemm_variant(1)
        want = ops.gemm(a, w, **kw).float()
        for v in (0, 1, 2, 3):
            ops.set_gemm_variant(v)
            try:
                got = ops.gemm(a, w, **kw).float()
                err = float((got - want).abs().max() / want.abs().max())
                assert err < 2e-2, (name, v, err)
                t = timeit(lambda: ops.gemm(a, w, **kw))
                line += f" | v{v} (kernel {raw('simseg_gemm_last_variant')}): {t:7.1f} us {2.0 * M * N * K / t / 1e6:5.0f} TF {by / t / 1e3:5.0f} GB/s"
            except Exception as e:      # noqa: BLE001
                line += f" | v{v}: {str(e)[:30]}"
        ops.set_gemm_variant(0)
        print(line, flush=True)
