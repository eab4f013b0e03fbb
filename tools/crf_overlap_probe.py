#!/usr/bin/env python
"""Round 6: do the DenseCRF kernels and the encoder's kernels actually run side by side on two streams?
One ViT-B/16 bf16 encoder pass over 256 windows of 512^2 on stream A, the CRF stage of a 64-window chunk (x4 = 256 windows) on the
high-priority stream B: each alone, then both enqueued together.  SIMSEG_GEMM_PP2_RESERVE=r leaves r CUs to stream B.
    python tools/crf_overlap_probe.py           (one process per reserve setting: the knob is read once)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SIMSEG_AMD_COMPUTE", "bf16")
import bench  # noqa: E402
from simseg_amd import ops, segpost  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", RANK="0", WORLD_SIZE="1")
    from simseg.models import PIPELINE
    cfg, build = bench.build_model("vit_base_patch16_224_in21k", 768, 512)
    torch.manual_seed(7)
    model = build(cfg.model.name, cfg, PIPELINE).to(dev).eval()
    W, S = 256, 512
    g = torch.Generator().manual_seed(3)
    images = torch.randn(W, 3, S, S, generator=g).to(dev)
    yy, xx = torch.meshgrid(torch.arange(S), torch.arange(S), indexing="ij")
    base = torch.stack([(xx * 255 // S), (yy * 255 // S), ((xx + yy) * 127 // S)], -1).float()
    u8 = (base[None] + 20 * torch.randn(8, S, S, 3, generator=g)).clamp(0, 255).to(torch.uint8).repeat(8, 1, 1, 1).contiguous().to(dev)      # 64 windows
    prob = torch.nn.functional.avg_pool2d(torch.rand(64, 2, S // 16, S // 16, generator=g), 3, 1, 1)
    prob = ((prob - prob.amin((2, 3), keepdim=True)) / (prob.amax((2, 3), keepdim=True) - prob.amin((2, 3), keepdim=True))).to(dev)
    up = prob.repeat_interleave(16, 2).repeat_interleave(16, 3).contiguous()
    (sa, _), sb = segpost._pipeline_streams(dev)

    def enc():
        with torch.no_grad():
            return model.forward_image_feature(images)

    def crf():
        for _ in range(4):
            ops.dense_crf(u8, up, **segpost.CRF_PARAMS)

    def timed(fa, fb):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        sa.wait_stream(torch.cuda.current_stream()); sb.wait_stream(torch.cuda.current_stream())
        if fa:
            with torch.cuda.stream(sa):
                fa()
        if fb:
            with torch.cuda.stream(sb):
                fb()
        torch.cuda.current_stream().wait_stream(sa); torch.cuda.current_stream().wait_stream(sb)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    for _ in range(2):
        timed(enc, crf)
    r = os.environ.get("SIMSEG_GEMM_PP2_RESERVE", "0")
    for _ in range(2):
        te, tc, tb = timed(enc, None), timed(None, crf), timed(enc, crf)
        print(f"reserve {r:>3}: encoder alone {te:7.2f} ms | CRF (4 x 64 windows, 2 maps) alone {tc:7.2f} ms | both streams {tb:7.2f} ms "
              f"(sum {te + tc:7.2f}, saved {te + tc - tb:6.2f})", flush=True)


if __name__ == "__main__":
    main()
