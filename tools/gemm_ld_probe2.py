#!/usr/bin/env python
"""Leading-dimension padding of the WEIGHT operand only (x . W^T forward: NT, W [out, in] with pitch in + pad; dgrad d . W: NN, the same
buffer as [K = out, N = in]) at realistic operand scales (activations ~N(0,1), weights ~N(0, 0.02^2)), interleaved rounds, median us."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simseg_amd import ops  # noqa: E402
from simseg_amd.lib import call, ptr  # noqa: E402
from simseg_amd.ops import stream  # noqa: E402

MV = 512 * 197
g = torch.Generator(device="cuda").manual_seed(0)
tot = {}
for name, kind, M, N, K in (("qkv fwd", "nt", MV, 2304, 768), ("proj fwd", "nt", MV, 768, 768), ("fc1 fwd", "nt", MV, 3072, 768), ("fc2 fwd", "nt", MV, 768, 3072),
                            ("qkv dgrad", "nn", MV, 768, 2304), ("proj dgrad", "nn", MV, 768, 768), ("fc1 dgrad", "nn", MV, 768, 3072), ("fc2 dgrad", "nn", MV, 3072, 768)):
    a = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    rows, cols = (N, K) if kind == "nt" else (K, N)          # the weight buffer [out, in]: NT reads it as [N, K], NN as [K, N]
    fns = {}
    for pad in (0, 64, 32):
        w = (torch.randn(rows, cols + pad, device="cuda", generator=g) * 0.02).bfloat16()
        fns[pad] = (lambda w=w, pad=pad: call("simseg_gemm", ptr(a), ptr(w), ptr(out), M, N, K, K, cols + pad, N, 1, 1, 0, int(kind == "nn"), 1.0, None, None, None, 0, 0,
                                              None, None, 0, 0, 0, 1, 0, 0.0, None, stream()))
    times = {p: [] for p in fns}
    for r in range(6):
        order = list(fns) if r % 2 == 0 else list(fns)[::-1]
        for p in order:
            for _ in range(3):
                fns[p]()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fns[p]()
            e1.record()
            torch.cuda.synchronize()
            times[p].append(e0.elapsed_time(e1) * 100)
    med = {p: statistics.median(t) for p, t in times.items()}
    for p in med:
        tot[p] = tot.get(p, 0.0) + med[p]
    print(f"{name:<11} {kind} M={M} N={N:4d} K={K:4d}: " + " | ".join(f"pitch +{p:<2d} {med[p]:7.1f} us ({100 * (med[p] / med[0] - 1):+5.1f} %)" for p in med), flush=True)
print("sum: " + " | ".join(f"+{p}: {t:.1f} us ({100 * (t / tot[0] - 1):+.1f} %)" for p, t in tot.items()))
