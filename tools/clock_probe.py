#!/usr/bin/env python
"""Shader clock and power while one GEMM shape runs back to back (rocm-smi sampled from a side thread).  argv: M N K seconds"""
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simseg_amd import ops  # noqa: E402

M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (100864, 768, 3072)
secs = float(sys.argv[4]) if len(sys.argv) > 4 else 6.0
a = torch.randn(M, K, device="cuda").bfloat16()
b = torch.randn(N, K, device="cuda").bfloat16()
samples = []
stop = False


def sample():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
            sclk = [l for l in out.splitlines() if "sclk" in l]
            pw = [l for l in out.splitlines() if "ower" in l and "W" in l]
            samples.append((time.time(), sclk[0].strip() if sclk else "?", pw[0].strip() if pw else "?"))
        except Exception as e:  # noqa: BLE001
            samples.append((time.time(), repr(e), ""))
        time.sleep(0.5)


th = threading.Thread(target=sample)
th.start()
time.sleep(1.5)
t0 = time.time()
n = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while time.time() - t0 < secs:
    for _ in range(50):
        ops.gemm(a, b)
    n += 50
    torch.cuda.synchronize()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
time.sleep(1.0)
stop = True
th.join()
print(f"{M}x{N}x{K}: {ms * 1e3:.1f} us per launch = {2.0 * M * N * K / ms / 1e9:.0f} TFLOP/s")
for t, s, p in samples:
    print(f"  t={t - t0:5.1f}s  {s}   {p}")
