#!/usr/bin/env python
"""How far ahead of the GPU does the host run in the training step?  (round 4)

For each variant of the text tower's row-map construction -
    index_ops      : the round-3 torch index kernels with their two host reads (SIMSEG_AMD_RAGGED_KERNEL=0)
    kernel         : one HIP kernel, the real-token count read back (one host read)
    kernel+lengths : one HIP kernel, packed row count from batch["caption_lengths"] (host numbers): no host read in the step
it prints
    wall      steady-state ms per step (K steps back to back, one sync at the end) - what bench.py times
    enqueue   ms the host needs to enqueue one whole step when nothing blocks it (GPU idle at the start of the step; syncs inside the
              step show up here as waiting for the GPU)
    fwd / bwd / opt   the same split by phase
Usage: python tools/step_host_lead.py [--steps 10] [--pairs 512]"""
import argparse
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--pairs", type=int, default=512)
    ap.add_argument("--variants", default="index_ops,kernel,kernel+lengths")
    args = ap.parse_args()
    os.environ["SIMSEG_AMD_COMPUTE"] = "bf16"
    import bench
    from simseg.models import PIPELINE
    from simseg.utils import logger
    from simseg_amd import towers
    from simseg_amd.optim import AdamW
    logger.STREAM = sys.stderr
    dev = torch.device("cuda", 0)
    cfg, build = bench.build_model("vit_base_patch16_224_in21k", 768, 224)
    torch.manual_seed(1234)
    model = build(cfg.model.name, cfg, PIPELINE).to(dev).train()
    opt = AdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=1e-3)
    batches = [bench.synthetic_batch(args.pairs, 224, 77, 30522, 1000 + 100 * i, dev) for i in range(4)]
    n = [0]

    def next_batch(lengths):
        b = batches[n[0] % 4]
        n[0] += 1
        out = {"image": b["image"], "input_ids": b["input_ids"].clone(), "attention_mask": b["attention_mask"].clone()}
        if lengths:
            out["caption_lengths"] = b["caption_lengths"]
        return out

    def step(lengths, stamps=None):
        opt.zero_grad(set_to_none=True)
        loss = model(next_batch(lengths))[0]["nce_loss"]
        if stamps is not None:
            stamps.append(time.perf_counter())
        loss.backward()
        if stamps is not None:
            stamps.append(time.perf_counter())
        opt.step()
        if stamps is not None:
            stamps.append(time.perf_counter())

    print(f"{'variant':<16} {'wall ms':>8} {'enqueue ms':>11} {'fwd':>7} {'bwd':>7} {'opt':>6}   (host-side ms per step; {args.pairs} pairs)")
    for v in args.variants.split(","):
        towers._RAGGED_KERNEL = v != "index_ops"
        lengths = v.endswith("+lengths")
        for _ in range(3):
            step(lengths)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(lengths)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / args.steps
        enq = [0.0, 0.0, 0.0]
        for _ in range(args.steps):
            torch.cuda.synchronize()
            st = [time.perf_counter()]
            step(lengths, st)
            for i in range(3):
                enq[i] += st[i + 1] - st[i]
        torch.cuda.synchronize()
        e = [x / args.steps * 1e3 for x in enq]
        print(f"{v:<16} {wall * 1e3:8.2f} {sum(e):11.2f} {e[0]:7.2f} {e[1]:7.2f} {e[2]:6.2f}", flush=True)


if __name__ == "__main__":
    main()
