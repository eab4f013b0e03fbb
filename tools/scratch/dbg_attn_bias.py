#!/usr/bin/env python
"""Debug: column sums of dK (the key-bias gradient, zero in exact arithmetic) from the HIP attention backward vs torch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from simseg_amd import ops  # noqa: E402


def ref_bwd(qkv, dout, H, mask):
    B, T, _ = qkv.shape
    x = qkv.float().requires_grad_(True)
    q, k, v = x.view(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = q @ k.transpose(-1, -2) * 0.125
    if mask is not None:
        s = s + (1.0 - mask[:, None, None, :].float()) * -10000.0
    o = (s.softmax(-1) @ v).transpose(1, 2).reshape(B, T, H * 64)
    o.backward(dout.float())
    return x.grad


for (B, T, H, masked, qs, ds) in ((6, 77, 12, True, 0.55, 1e-3), (6, 77, 12, False, 0.55, 1e-3), (4, 197, 12, False, 0.55, 1e-3),
                                  (2, 77, 2, True, 1.2, 1.0), (6, 25, 12, True, 0.55, 1e-3), (6, 64, 12, False, 0.55, 1e-3), (6, 96, 12, False, 0.55, 1e-3)):
    g = torch.Generator(device="cuda").manual_seed(T)
    qkv = (torch.randn(B, T, 3 * H * 64, device="cuda", generator=g) * qs).bfloat16()
    dout = (torch.randn(B, T, H * 64, device="cuda", generator=g) * ds).bfloat16()
    mask = None
    if masked:
        mask = torch.zeros(B, T, dtype=torch.long, device="cuda")
        lens = torch.randint(8, T + 1, (B,), generator=torch.Generator().manual_seed(1))
        for b in range(B):
            mask[b, :lens[b]] = 1
        dout = dout * mask[:, :, None].bfloat16()
    out, lse = ops.attention_fwd(qkv, H, mask, save_lse=True)
    dqkv = ops.attention_bwd(qkv, out, dout, lse, H, mask).float()
    ref = ref_bwd(qkv, dout, H, mask)
    d = dqkv.view(B, T, 3, H * 64)
    r = ref.view(B, T, 3, H * 64)
    names = ["dQ", "dK", "dV"]
    msg = [f"B={B} T={T} H={H} masked={masked}"]
    for i in range(3):
        err = (d[:, :, i] - r[:, :, i])
        cs_o, cs_r = d[:, :, i].sum((0, 1)), r[:, :, i].sum((0, 1))
        msg.append(f"{names[i]}: rel err {float(err.norm() / r[:, :, i].norm()):.2e} colsum ours {float(cs_o.norm()):.3e} ref {float(cs_r.norm()):.3e} "
                   f"colsum err {float((cs_o - cs_r).norm()):.3e}")
    # per-key-row error profile of dK for batch 0
    e = (d[:, :, 1] - r[:, :, 1]).view(B, T, H, 64).norm(dim=-1)      # [B,T,H]
    msg.append("dK row err by key index (mean over b,h): " + " ".join(f"{float(e[:, t].mean()):.1e}" for t in range(0, T, max(1, T // 16))))
    print("\n  ".join(msg))
