#!/usr/bin/env python
"""bf16 attention backward: the single-kernel form against the two resident passes (SIMSEG attention variant 3), same box,
at the ViT-B training shape and the BERT caption shape; also the largest element difference between the two."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simseg_amd import ops  # noqa: E402


def timeit(fn, iters=20):
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
    g = torch.Generator(device="cuda").manual_seed(0)
    H = 12
    for name, B, T, masked, drop in (("vitb_224", 512, 197, False, 0.0), ("bert_77", 512, 77, True, 0.0), ("bert_77_dropout", 512, 77, True, 0.1),
                                      ("bert_77_packed", 512, 77, True, 0.1), ("t224", 64, 224, False, 0.0), ("t33", 64, 33, False, 0.0)):
        qkv = torch.randn(B, T, 3 * H * 64, device="cuda", generator=g).to(torch.bfloat16)
        mask = None
        if masked:
            lens = torch.randint(5, T + 1, (B,), device="cuda", generator=g)
            mask = (torch.arange(T, device="cuda")[None] < lens[:, None]).long()
        kw = dict(drop_seed=1234, drop_p=drop, skip_padded_rows=name.endswith("packed"))
        o, lse = ops.attention_fwd(qkv, H, mask, save_lse=True, **kw)
        do = torch.randn_like(o)
        if mask is not None:
            do = do * mask[:, :, None].to(do.dtype)       # the caller contract of skip_padded_rows: no gradient arrives on padded rows
        cs = torch.zeros(3 * H * 64, device="cuda")
        timeit(lambda: ops.attention_bwd(qkv, o, do, lse, H, mask, **kw), iters=60)       # (clock ramp after the idle set-up phase: the first ~30 ms run slower)
        us = timeit(lambda: ops.attention_bwd(qkv, o, do, lse, H, mask, colsum=cs, **kw))
        print(f"{name} B={B} T={T} one kernel + q/k/v bias gradient: {us:.1f} us", flush=True)
        res = {}
        for v in (0, 3):
            ops.set_attention_variant(v)
            res[v] = ops.attention_bwd(qkv, o, do, lse, H, mask, **kw).float()
            us = timeit(lambda: ops.attention_bwd(qkv, o, do, lse, H, mask, **kw))
            print(f"{name} B={B} T={T} variant={v}: {us:.1f} us", flush=True)
        ops.set_attention_variant(0)
        if mask is not None:
            res = {k: r * mask[:, :, None] for k, r in res.items()}
        d = (res[0] - res[3]).abs()
        print(f"   max |one - two| = {d.max().item():.3e} (max |two| = {res[3].abs().max().item():.3e}), elements differing: {(d > 0).float().mean().item():.4f}")


if __name__ == "__main__":
    main()
