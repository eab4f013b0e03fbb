"""Round 6: phase timeline of the 64-queries-per-wave forward (simseg_debug_attention_timeline on the w64 kernel): cycle sums per phase of
wave 0 of every block, block lifetimes and co-residency.  argv: attention variants to run (0 = two query blocks per wave, 6 = one)."""
import collections
import os
import sys
import torch
sys.path.insert(0, "/root/repo")
from simseg_amd import ops
from simseg_amd.lib import call, ptr, stream

variants = [int(a) for a in sys.argv[1:]] or [0]
for variant in variants:
    rb = 256 if variant == 0 else 128
    for (B, T, H) in ((16, 1025, 12), (256, 1025, 12)):
        qkv = torch.randn(B, T, 3 * H * 64, device="cuda").bfloat16()
        if os.environ.get("ZERO") == "1":
            qkv.zero_()
        out = torch.empty(B, T, H * 64, device="cuda", dtype=torch.bfloat16)
        left = T % 256
        gxm = T // 256 + (1 if left > 128 else 0)
        gx = (gxm + ((left + 63) // 64 if 0 < left <= 128 else 0)) if variant == 0 else (T + rb - 1) // rb
        nblk = gx * B * H
        dbg = torch.zeros(64 + 12 * nblk, device="cuda", dtype=torch.int64)
        ops.set_attention_variant(variant)
        for _ in range(3):
            call("simseg_debug_attention_timeline", ptr(qkv), ptr(out), None, ptr(dbg), B, T, H, stream())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        call("simseg_debug_attention_timeline", ptr(qkv), ptr(out), None, ptr(dbg), B, T, H, stream())
        e1.record()
        torch.cuda.synchronize()
        ops.set_attention_variant(0)
        ms = e0.elapsed_time(e1)
        d = dbg.cpu().tolist()
        nt = (T - 1) // 64
        print(f"variant {variant} B={B} T={T} H={H}: {nblk} blocks of {rb} rows, kernel {ms * 1e3:.1f} us (instrumented)")
        rec = torch.tensor(d[64:]).view(nblk, 12)
        ph = rec[:, 4:11].float()
        main = torch.ones(nblk, dtype=torch.bool).view(B * H, gx)
        if variant == 0:
            main[:, gxm:] = False
        for name, sel in (("main", main.view(-1)),):
            pm = ph[sel].mean(0)
            life = (rec[sel, 1] - rec[sel, 0]).float().mean()
            print(f"   {name} blocks, wave 0: prologue {pm[0]:.0f}; per tile: tile wait {pm[5] / nt:.0f}, barrier {pm[6] / nt:.0f}, copies issued {pm[1] / nt:.0f}, "
                  f"phases 1+2 {pm[2] / nt:.0f}, phase 3 {pm[3] / nt:.0f}, phase 4 {pm[4] / nt:.0f} = {(pm[1:].sum()) / nt:.0f}; lifetime {life:.0f}")
        gs, ge = rec[rec[:, 1] > 0, 0].min().item(), rec[:, 1].max().item()
        print(f"   whole launch: {ge - gs} counter ticks in {ms * 1e3:.1f} us = {(ge - gs) / (ms * 1e3):.1f} ticks/us (the counter __builtin_readcyclecounter reads)")
        hw, xcc = rec[:, 2], rec[:, 3] & 0xF
        key = list(zip(xcc.tolist(), ((hw >> 13) & 7).tolist(), ((hw >> 12) & 1).tolist(), ((hw >> 8) & 15).tolist()))
        per = collections.defaultdict(list)
        for k, (s, e) in zip(key, rec[:, :2].tolist()):
            if e > 0:
                per[k].append((s, e))
        # residency histogram per CU: share of the CU's busy span with 0 / 1 / 2 / more blocks resident (wave 0's first to last instruction),
        # and the gap between a block's end and the next block start on that CU
        hist, gaps = collections.Counter(), []
        for v in per.values():
            ev = sorted([(s, 1) for s, _ in v] + [(e, -1) for _, e in v])
            lo, hi = ev[0][0], ev[-1][0]
            a, b = lo + (hi - lo) * 0.1, lo + (hi - lo) * 0.9           # steady state: the middle 80 % of the span
            n, t_prev = 0, ev[0][0]
            for t, dlt in ev:
                w = max(0, min(t, b) - max(t_prev, a))
                hist[min(n, 3)] += w
                n += dlt
                t_prev = t
            ends = sorted(e for _, e in v)
            starts = sorted(s for s, _ in v)
            import bisect
            for e in ends:
                j = bisect.bisect_left(starts, e)
                if j < len(starts) and a < e < b:
                    gaps.append(starts[j] - e)
        tot = sum(hist.values())
        gaps.sort()
        print("   steady state (middle 80 % of every CU's span): share of time with 0 / 1 / 2 / >2 blocks resident: "
              + " / ".join(f"{hist[i] / tot:.3f}" for i in range(4))
              + f"; end of a block -> next block start on that CU: median {gaps[len(gaps) // 2]} ticks, mean {sum(gaps) / len(gaps):.0f}, p90 {gaps[len(gaps) * 9 // 10]}")
        conc = [sum(e - s for s, e in v) / max(1, max(e for _, e in v) - min(s for s, _ in v)) for v in per.values()]
        span = [max(e for _, e in v) - min(s for s, _ in v) for v in per.values()]
        print(f"   distinct CUs {len(per)}, mean concurrent blocks per CU {sum(conc) / len(conc):.2f}, blocks per CU {nblk / len(per):.1f}, "
              f"CU busy span mean {sum(span) / len(span):.0f} ticks => {sum(span) / len(span) / (ms * 1e3):.0f} ticks/us")
