#!/usr/bin/env python
"""Glue census of ONE default training step (512 pairs, bf16, packed captions with host-side lengths): every ops.cast, torch.zeros /
zeros_like / Tensor.zero_ / torch.empty above 1 MB with its call site, so that passes over whole activations that are not part of a
GEMM / attention / LayerNorm kernel can be named and removed.  Then N plain steps for a rocprofv3 --kernel-trace around this process
(tools/rocpd_stats.py divides by the step count).
Usage: python tools/step_census.py [--steps 6]"""
import argparse
import collections
import os
import sys
import traceback

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "simseg" in fr.filename and not fr.filename.endswith("ops.py") and "step_census" not in fr.filename:
            return f"{os.path.basename(fr.filename)}:{fr.lineno}"
    return "?"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--pairs", type=int, default=512)
    args = ap.parse_args()
    os.environ["SIMSEG_AMD_COMPUTE"] = "bf16"
    import bench
    from simseg.models import PIPELINE
    from simseg.utils import logger
    from simseg_amd import ops
    from simseg_amd.optim import AdamW
    logger.STREAM = sys.stderr
    dev = torch.device("cuda", 0)
    cfg, build = bench.build_model("vit_base_patch16_224_in21k", 768, 224)
    torch.manual_seed(1234)
    model = build(cfg.model.name, cfg, PIPELINE).to(dev).train()
    opt = AdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=1e-3)
    batches = [bench.synthetic_batch(args.pairs, 224, 77, 30522, 1000 + 100 * i, dev) for i in range(4)]
    n = [0]

    def step():
        b = batches[n[0] % 4]
        n[0] += 1
        opt.zero_grad(set_to_none=True)
        loss = model({"image": b["image"], "input_ids": b["input_ids"].clone(), "attention_mask": b["attention_mask"].clone(),
                      "caption_lengths": b["caption_lengths"]})[0]["nce_loss"]
        loss.backward()
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    seen = collections.Counter()
    byts = collections.Counter()
    real = dict(cast=ops.cast, zeros=torch.zeros, zeros_like=torch.zeros_like, zero_=torch.Tensor.zero_)

    def note(kind, t):
        nb = t.numel() * t.element_size()
        if nb >= (1 << 20):
            k = (kind, site(), tuple(t.shape), str(t.dtype).replace("torch.", ""))
            seen[k] += 1
            byts[k] += nb

    def cast(x, *a, **k):
        note("cast", x)
        return real["cast"](x, *a, **k)

    def zeros(*a, **k):
        t = real["zeros"](*a, **k)
        note("zeros", t)
        return t

    def zeros_like(x, *a, **k):
        t = real["zeros_like"](x, *a, **k)
        note("zeros_like", t)
        return t

    def zero_(self):
        note("zero_", self)
        return real["zero_"](self)

    ops.cast, torch.zeros, torch.zeros_like, torch.Tensor.zero_ = cast, zeros, zeros_like, zero_
    try:
        step()
        torch.cuda.synchronize()
    finally:
        ops.cast, torch.zeros, torch.zeros_like, torch.Tensor.zero_ = real["cast"], real["zeros"], real["zeros_like"], real["zero_"]
    print("# passes >= 1 MB in one step: kind, call site, shape, dtype, calls, MB per step")
    for k, c in sorted(seen.items(), key=lambda kv: -byts[kv[0]]):
        print(f"{k[0]:<10} {k[1]:<22} {str(k[2]):<22} {k[3]:<9} x{c:<3} {byts[k] / 1e6:9.1f} MB")
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    print(f"# then {args.steps} plain steps")


if __name__ == "__main__":
    main()
