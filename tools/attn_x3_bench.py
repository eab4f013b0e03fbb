#!/usr/bin/env python
"""Exact-mode attention forward: the fp32 MFMA kernel against the split-bf16 form (simseg_attention_fwd_x3, split pass included)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simseg_amd import ops
for B, T, H in ((63, 1025, 12), (16, 1025, 12), (64, 325, 6), (512, 197, 12)):
    qkv = torch.randn(B, T, 3 * H * 64, device="cuda")
    res = {}
    for name, fn in (("fp32", lambda: ops.attention_fwd(qkv, H, None, scale=0.125)), ("x3", lambda: ops.attention_fwd_x3(qkv, H, scale=0.125))):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record(); torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 10
    planes = ops.split_bf16x3(qkv.view(B * T, -1), planes=True)
    out = torch.empty(B, T, H * 64, device="cuda")
    from simseg_amd.lib import call, ptr, stream
    for name, fn in (("split", lambda: ops.split_bf16x3(qkv.view(B * T, -1), planes=True)),
                     ("kernel", lambda: call("simseg_attention_fwd_x3", ptr(planes), B * T * 3 * H * 64, ptr(out), B, T, H, 0.125, stream()))):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record(); torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 10
    if T == 1025:
        for dbg, what in ((8, "no copies in the loop"), (16, "no tile barrier"), (24, "neither"), (7, "barriers / copies only"), (15, "barriers only"), (31, "empty loop")):
            ops.set_attention_variant(200 + dbg)
            fn = lambda: call("simseg_attention_fwd_x3", ptr(planes), B * T * 3 * H * 64, ptr(out), B, T, H, 0.125, stream())
            fn(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record(); torch.cuda.synchronize()
            print(f"   ablation {what}: {e0.elapsed_time(e1) / 10:.3f} ms")
        ops.set_attention_variant(0)
    fl = 4.0 * T * T * 64 * B * H
    print(f"B={B} T={T} H={H}: fp32 kernel {res['fp32']:.3f} ms ({fl / res['fp32'] / 1e9:.0f} TF)   split-bf16 {res['x3']:.3f} ms ({fl / res['x3'] / 1e9:.0f} TF fp32-equivalent) = split pass {res['split']:.3f} + kernel {res['kernel']:.3f} ms ({6 * fl / res['kernel'] / 1e9:.0f} TF of bf16 MFMA work)")
