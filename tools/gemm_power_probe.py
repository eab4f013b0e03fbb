#!/usr/bin/env python
"""Is the GEMM K loop bound by operand delivery or by the power envelope?  The same launches (same bytes moved, same instructions) on
Gaussian operands, on operands with few distinct values, and on zeros: a delivery-bound kernel does not care what the bits are; a
power-capped one speeds up as the multiplier arrays toggle less.  Clock / power sampled from sysfs during each run."""
import glob
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simseg_amd import ops  # noqa: E402


def sampler(stop, out):
    hw = glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average") + glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input")
    fq = glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")
    while not stop.is_set():
        try:
            w = max(int(open(p).read()) for p in hw) / 1e6 if hw else float("nan")
            mhz = float("nan")
            for p in fq:
                for line in open(p):
                    if line.strip().endswith("*"):
                        import re
                        mhz = max(mhz if mhz == mhz else 0.0, float(re.search(r"(\d+)\s*[Mm][Hh]z", line).group(1)))      # (the busy card of a multi-GPU node)
            out.append((w, mhz))
        except Exception:       # noqa: BLE001
            pass
        time.sleep(0.02)


def run(name, a, b, kw, flops, secs=1.5):
    for _ in range(5):
        ops.gemm(a, b, **kw)
    torch.cuda.synchronize()
    stop, samples = threading.Event(), []
    th = threading.Thread(target=sampler, args=(stop, samples))
    th.start()
    n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    while time.perf_counter() - t0 < secs:
        for _ in range(20):
            ops.gemm(a, b, **kw)
        n += 20
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    stop.set(); th.join()
    ms = e0.elapsed_time(e1) / n
    w = sum(s[0] for s in samples) / max(1, len(samples))
    mhz = sum(s[1] for s in samples) / max(1, len(samples))
    print(f"  {name:<34} {ms * 1e3:8.1f} us  {flops / ms / 1e9:7.0f} TFLOP/s   {w:6.0f} W  {mhz:6.0f} MHz", flush=True)


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    for kind, M, N, K in (("nt", 100864, 3072, 768), ("nn", 100864, 768, 3072), ("tn", 3072, 768, 100864)):
        print(f"{kind} M={M} N={N} K={K}")
        ta, tb = kind == "tn", kind in ("nn", "tn")
        sa = (K, M) if ta else (M, K)
        sb = (K, N) if tb else (N, K)
        kw = dict(trans_a=ta, trans_b=tb)
        if kind == "tn":
            kw.update(out=torch.zeros(M, N, device="cuda"), accumulate=True, splitk=max(1, 256 // (((M + 255) // 256) * ((N + 255) // 256))))
        else:
            kw.update(out=torch.empty(M, N, device="cuda", dtype=torch.bfloat16))
        fl = 2.0 * M * N * K
        for name, mk in (("gaussian", lambda s: torch.randn(s, device="cuda", generator=g)),
                         ("sign only (+-1)", lambda s: torch.randn(s, device="cuda", generator=g).sign()),
                         ("gaussian, weights x 0.02", None),
                         ("ones", lambda s: torch.ones(s, device="cuda")),
                         ("zeros", lambda s: torch.zeros(s, device="cuda"))):
            if mk is None:       # realistic scales: the weight operand (B in nt / nn) ~N(0, 0.02^2); the weight gradient has two activation operands
                a = torch.randn(sa, device="cuda", generator=g).bfloat16()
                b = (torch.randn(sb, device="cuda", generator=g) * (1.0 if kind == "tn" else 0.02)).bfloat16()
            else:
                a, b = mk(sa).bfloat16(), mk(sb).bfloat16()
            run(name, a, b, kw, fl)
            del a, b


if __name__ == "__main__":
    main()
