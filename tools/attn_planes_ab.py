#!/usr/bin/env python
"""bf16 attention at the training shapes on the packed projection rows [B, T, 3, H, 64] against the plane-major operands [3 H][B T][64]
(simseg_attention_{fwd,bwd}_planes): same kernels, same bits, only the addresses differ.  Same box, interleaved."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simseg_amd import ops  # noqa: E402


def timeit(fn, iters=30):
    for _ in range(5):
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
    for name, B, T, ragged, drop in (("vitb_224", 512, 197, False, 0.0), ("bert_77_ragged", 512, 77, True, 0.1), ("t224", 256, 224, False, 0.0)):
        if ragged:
            lens = torch.randint(8, T + 1, (B,), device="cuda", generator=g)
            rs = torch.zeros(B + 1, dtype=torch.int32, device="cuda")
            rs[1:] = lens.cumsum(0)
            n = int(rs[-1])
            rows = (n + 255) // 256 * 256
        else:
            rs, n, rows = None, B * T, B * T
        qkv = torch.randn(rows, 3 * H * 64, device="cuda", generator=g).to(torch.bfloat16)
        qkvp = qkv.view(rows, 3 * H, 64).permute(1, 0, 2).contiguous()
        kw = dict(drop_seed=77, drop_p=drop)
        if ragged:
            fa = lambda: ops.attention_fwd_rows(qkv, H, rs, T, save_lse=True, n_real=n, **kw)
        else:
            fa = lambda: ops.attention_fwd(qkv.view(B, T, -1), H, None, save_lse=True)
        fb = lambda: ops.attention_fwd_planes(qkvp, H, B, T, rs, save_lse=True, n_real=n, **kw)
        oa, la = fa()
        ob, lb = fb()
        oa = oa.reshape(rows, -1)
        print(f"{name}: fwd out equal {torch.equal(oa[:n], ob[:n])}, lse max diff {(la - lb).abs().max().item() if not ragged else float('nan'):.1e}")
        do = torch.randn_like(ob)
        if ragged:
            ba = lambda: ops.attention_bwd_rows(qkv, oa, do, la, H, rs, T, n_real=n, **kw)
        else:
            ba = lambda: ops.attention_bwd(qkv.view(B, T, -1), oa.view(B, T, -1), do.view(B, T, -1), la, H, None)
        bb = lambda: ops.attention_bwd_planes(qkvp, ob, do, lb, H, B, T, rs, n_real=n, **kw)
        da = ba().reshape(rows, 3 * H, 64)
        db = bb().permute(1, 0, 2)
        print(f"{name}: bwd dqkv equal {torch.equal(da[:n], db[:n])}")
        for rnd in range(3):
            t = [timeit(f) for f in (fa, fb, ba, bb)]
            print(f"  {name} B={B} T={T}: fwd rows {t[0]:.1f} us  planes {t[1]:.1f} us | bwd rows {t[2]:.1f} us  planes {t[3]:.1f} us", flush=True)
        for v, label in ((103, "no tile loop"), (102, "no copies")):
            ops.set_attention_variant(v)
            t = [timeit(f) for f in (fa, fb)]
            print(f"  fwd ablation ({label}): rows {t[0]:.1f} us  planes {t[1]:.1f} us")
        ops.set_attention_variant(0)


if __name__ == "__main__":
    main()
