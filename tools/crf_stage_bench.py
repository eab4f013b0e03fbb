#!/usr/bin/env python
"""The DenseCRF stage of the seg evaluation ALONE (nothing else on the GPU): 63 windows of 512^2, ~2 visited candidate maps per window, the
chunked device CRF of segpost.crf_masks (tools/seg_evaluation.py:31-54, :153).  Prints ms per batch (events around the whole stage), the GPU
time inside the simseg_dense_crf calls (events around each call) and what is left (host loop, candidate-table read, gather / scatter glue).
    python tools/crf_stage_bench.py [windows] [size]          rocprofv3 --kernel-trace -- python tools/crf_stage_bench.py   (per-kernel)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simseg_amd import ops, segpost  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 63
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(3)
    n = S // 16
    yy, xx = torch.meshgrid(torch.arange(S), torch.arange(S), indexing="ij")
    base = torch.stack([(xx * 255 // S), (yy * 255 // S), ((xx + yy) * 127 // S)], -1).float()
    u8 = (base[None] + 20 * torch.randn(min(B, 8), S, S, 3, generator=g)).clamp(0, 255).to(torch.uint8)
    u8 = u8.repeat((B + 7) // 8, 1, 1, 1)[:B].contiguous().to(dev)
    # candidate maps with structure (blobs), 2 visited slots per window on average (0..4)
    K = 5
    prob = torch.rand(B, K, n, n, generator=g)
    prob = torch.nn.functional.avg_pool2d(prob, 3, 1, 1)
    prob = ((prob - prob.amin((2, 3), keepdim=True)) / (prob.amax((2, 3), keepdim=True) - prob.amin((2, 3), keepdim=True))).to(dev)
    nv = torch.randint(1, 4, (B,), generator=g)
    cand = torch.full((B, K), -1, dtype=torch.int32)
    for b in range(B):
        cand[b, :int(nv[b])] = torch.arange(1, int(nv[b]) + 1, dtype=torch.int32)
    cand = cand.to(dev)
    times = []
    real = ops.dense_crf

    def timed(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = real(*a, **k)
        e1.record()
        times.append((e0, e1))
        return r

    ops.dense_crf = timed
    for it in range(4):
        times.clear()
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s0.record()
        m = segpost.crf_masks(prob, cand, u8, scale=16)
        s1.record()
        torch.cuda.synchronize()
        inside = sum(a.elapsed_time(b) for a, b in times)
        print(f"batch of {B} x {S}^2, {float(nv.float().mean()):.2f} visited maps per window: stage {s0.elapsed_time(s1):7.2f} ms, inside {len(times)} simseg_dense_crf "
              f"calls {inside:7.2f} ms, rest {s0.elapsed_time(s1) - inside:6.2f} ms   (mask pixels set: {int((m > 0).sum())})", flush=True)


if __name__ == "__main__":
    main()
