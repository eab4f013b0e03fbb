"""Debug: per-phase cycle counters and per-block start/end/CU records of the bf16 attention forward (MI355X only)."""
import os, sys, torch, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simseg_amd.lib import call, ptr, stream


def waves(q32):          # attn_waves_per_block in simseg_amd/csrc/attn.hip
    return q32 if q32 <= 4 else 4


for (B, T, H) in ((1, 197, 1), (512, 197, 12), (16, 1025, 12)):
    qkv = torch.randn(B, T, 3 * H * 64, device="cuda").bfloat16()
    out = torch.empty(B, T, H * 64, device="cuda", dtype=torch.bfloat16)
    lse = torch.empty(B, H, T, device="cuda")
    q32 = (T + 31) // 32
    nblk = (q32 + waves(q32) - 1) // waves(q32) * B * H
    dbg = torch.zeros(8 + 4 * nblk, device="cuda", dtype=torch.int64)
    for _ in range(3):
        call("simseg_debug_attention_timeline", ptr(qkv), ptr(out), ptr(lse), ptr(dbg), B, T, H, stream())
    torch.cuda.synchronize()
    d = dbg.cpu().tolist()
    nt = (T + 63) // 64
    print(f"B={B} T={T} H={H}: blocks {nblk}, block(0,0) lifetime {d[5]} ticks; first-tile staging {d[0]}, per K/V tile: S {d[1] // nt}, "
          f"softmax {d[2] // nt}, PV {d[3] // nt}, commit+sync {d[4] // nt} (cycle-counter ticks)")
    rec = torch.tensor(d[8:]).view(nblk, 4)
    hw, xcc = rec[:, 2], rec[:, 3] & 0xF
    cu = (hw >> 8) & 0xF
    sh = (hw >> 12) & 0x1
    se = (hw >> 13) & 0x7
    key = (xcc * 64 + se * 8 + sh * 16 * 0 + cu).tolist()   # CU identity within the chip (xcc, se, cu)
    key = [(int(x), int(a), int(b_), int(c)) for x, a, b_, c in zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist())]
    per = collections.defaultdict(list)
    for k, (s, e) in zip(key, rec[:, :2].tolist()):
        per[k].append((s, e))
    life = (rec[:, 1] - rec[:, 0]).float()
    print(f"   distinct CUs seen {len(per)}; block lifetime ticks: mean {life.mean():.0f} p10 {life.quantile(0.1):.0f} p90 {life.quantile(0.9):.0f}")
    # average concurrency per CU = sum of lifetimes / (last end - first start) on that CU
    conc, spans = [], []
    for k, v in per.items():
        s0 = min(s for s, _ in v); e1 = max(e for _, e in v)
        conc.append(sum(e - s for s, e in v) / max(1, e1 - s0)); spans.append(e1 - s0)
    print(f"   mean concurrent blocks per CU {sum(conc) / len(conc):.2f}; CU busy span ticks mean {sum(spans) / len(spans):.0f} max {max(spans)}; "
          f"blocks per CU mean {nblk / len(per):.1f}")
