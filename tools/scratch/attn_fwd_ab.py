#!/usr/bin/env python
"""bf16 attention forward at the ViT-B training shape: the persistent loader-wave kernel (variant 4, opt-in) and its ablations against the
one-block-per-head resident kernel (variant 0 = auto) - same inputs, interleaved rounds, outputs compared.   usage: attn_fwd_ab.py [B=512] [T=197] [H=12]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from simseg_amd import ops  # noqa: E402


def timed(fn, iters=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 197
    H = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn(B, T, 3 * H * 64, device="cuda", generator=g).bfloat16()
    outs = {}
    for v in (4, 0):
        ops.set_attention_variant(v)
        outs[v] = ops.attention_fwd(qkv, H, None, scale=0.125, save_lse=True)
    torch.cuda.synchronize()
    d = float((outs[4][0].float() - outs[0][0].float()).abs().max())
    dl = float((outs[4][1] - outs[0][1]).abs().max())
    print(f"B={B} T={T} H={H}: max |out(persistent) - out(resident)| = {d:.3e}, lse {dl:.3e}")
    by = 2.0 * B * T * H * 64 * 4
    for rnd in range(3):
        row = []
        for v, name in ((4, "persistent"), (41, "pers: no tile loop"), (42, "pers: no copies"), (43, "pers: no stores"), (0, "resident (default)")):
            ops.set_attention_variant(v)
            us = timed(lambda: ops.attention_fwd(qkv, H, None, scale=0.125, save_lse=True))
            row.append(f"{name} {us:7.1f} us")
        print("  " + "   ".join(row), flush=True)
    ops.set_attention_variant(0)


if __name__ == "__main__":
    main()
