#!/usr/bin/env python
"""Shader clock / power sampled by rocm-smi while bench.py's training step loop runs in this process."""
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
samples, stop = [], False


def sample():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
            sclk = [l.split("(")[-1].strip(")") for l in out.splitlines() if "sclk" in l]
            pw = [l.split(":")[-1].strip() for l in out.splitlines() if "ower" in l and "W" in l]
            samples.append((time.time(), sclk[0] if sclk else "?", pw[0] if pw else "?"))
        except Exception as e:  # noqa: BLE001
            samples.append((time.time(), repr(e), ""))
        time.sleep(0.5)


th = threading.Thread(target=sample)
th.start()
sys.argv = ["bench.py", "--steps", "40", "--warmup", "3", "--no-seg", "--no-cpu-baseline"]
t0 = time.time()
import bench  # noqa: E402,F401

bench.main() if hasattr(bench, "main") else None
stop = True
th.join()
for t, s, p in samples:
    print(f"  t={t - t0:5.1f}s  sclk {s}   power {p} W")
