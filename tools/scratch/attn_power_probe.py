"""Round 6: is the long-sequence attention forward bound by instruction issue or by the power envelope?  The same launch (B = 256, T = 1025, 12 heads:
same instructions, same bytes) on Gaussian operands, on operands with few distinct values and on zeros; clock and package power sampled from sysfs
during each run (bench.ClockSampler: the card of this process's HIP device).  A kernel bound by issue slots does not care what the bits are."""
import os
import sys
import threading
import time

import torch

sys.path.insert(0, "/root/repo")
from simseg_amd import ops  # noqa: E402
from bench import ClockSampler  # noqa: E402

_pr = torch.cuda.get_device_properties(0)
BDF = f"{_pr.pci_domain_id:04x}:{_pr.pci_bus_id:02x}:{_pr.pci_device_id:02x}.0"


def run(name, fn, flops, secs=2.0):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    cs = ClockSampler(period=0.05, bdf=BDF)
    time.sleep(0.1)
    cs.mark()
    n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    while time.perf_counter() - t0 < secs:
        for _ in range(10):
            fn()
        n += 10
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    c = cs.stop() or {}
    ms = e0.elapsed_time(e1) / n
    w, mhz = c.get("power_w_avg") or float("nan"), c.get("sclk_mhz_avg") or float("nan")
    print(f"  {name:<44} {ms * 1e3:8.1f} us  {flops / ms / 1e9:7.0f} TFLOP/s  {flops / ms / 1e9 / 2500:.3f} of peak  {w:6.0f} W  {mhz:6.0f} MHz"
          f"  = {flops / ms / 1e9 / (2500 * mhz / 2400):.3f} of the matrix pipe at that clock", flush=True)


def main():
    H = 12
    c = ops.attention_qscale(0.125)
    for dtype in (torch.bfloat16, torch.float16):
        for (B, T) in ((256, 1025), (64, 2305)):
            print(f"{str(dtype)[6:]} B={B} T={T} H={H}")
            fl = 4.0 * B * H * T * T * 64
            g = torch.Generator(device="cuda").manual_seed(0)
            base = torch.randn(B, T, 3, H * 64, device="cuda", generator=g)
            for name, mk in (("gaussian (q pre-scaled)", lambda x: x),
                             ("sign only (+-1), q = +-c", lambda x: x.sign()),
                             ("q = 0 (all P = 1), k / v gaussian", lambda x: torch.cat([x[:, :, :1] * 0, x[:, :, 1:]], 2)),
                             ("v = 0, q / k gaussian", lambda x: torch.cat([x[:, :, :2], x[:, :, 2:] * 0], 2)),
                             ("zeros", lambda x: x * 0)):
                x = mk(base).clone()
                x[:, :, 0] *= c
                qkv = x.view(B, T, 3 * H * 64).to(dtype)
                run(name, lambda: ops.attention_fwd_qscaled(qkv, H), fl)
            qkv = base.view(B, T, 3 * H * 64).to(dtype)
            ops.set_attention_variant(1)
            run("ring kernel (round 5), gaussian", lambda: ops.attention_fwd(qkv, H, None, scale=0.125), fl)
            run("ring kernel (round 5), zeros", lambda: ops.attention_fwd(qkv * 0, H, None, scale=0.125), fl)
            ops.set_attention_variant(0)


main()
