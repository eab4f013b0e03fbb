#!/usr/bin/env python
"""Same-box comparison of the training step in the two 16-bit types and with / without a live GradScaler:
    bf16 | fp16 without a scaler (kernel flavour only) | fp16 + torch.amp.GradScaler | fp16 + simseg_amd.optim.GradScaler
ms per step over --steps steps each, two rounds (interleaved)."""
import argparse
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    args = ap.parse_args()
    import bench
    from simseg.models import PIPELINE
    from simseg.utils import logger
    from simseg_amd.optim import AdamW, GradScaler
    logger.STREAM = sys.stderr
    dev = torch.device("cuda", 0)
    os.environ["SIMSEG_AMD_COMPUTE"] = "bf16"
    cfg, build = bench.build_model("vit_base_patch16_224_in21k", 768, 224)
    torch.manual_seed(1234)
    model = build(cfg.model.name, cfg, PIPELINE).to(dev).train()
    opt = AdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=1e-3)
    batches = [bench.synthetic_batch(512, 224, 77, 30522, 1000 + 100 * i, dev) for i in range(4)]
    n = [0]

    def nb():
        b = batches[n[0] % 4]
        n[0] += 1
        return {"image": b["image"], "input_ids": b["input_ids"].clone(), "attention_mask": b["attention_mask"].clone(), "caption_lengths": b["caption_lengths"]}

    def run(kind):
        half = torch.bfloat16 if kind == "bf16" else torch.float16
        os.environ["SIMSEG_AMD_COMPUTE"] = "bf16" if kind == "bf16" else "fp16"
        if opt.half_dtype != half:
            opt.half_dtype = half
            opt._plans.clear()
        scaler = {"fp16+torch": torch.amp.GradScaler("cuda"), "fp16+ours": GradScaler("cuda")}.get(kind)

        def step():
            opt.zero_grad(set_to_none=True)
            loss = model(nb())[0]["nce_loss"]
            if scaler is None:
                loss.backward()
                opt.step()
            else:
                scaler.scale(loss).backward()
                scaler.step(opt)
                scaler.update()
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        clocks = bench.ClockSampler(period=0.1, bdf=bdf)
        time.sleep(0.3)
        clocks.mark()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        c = clocks.stop() or {}
        return ms, c.get("sclk_mhz_avg"), c.get("power_w_avg")

    pr = torch.cuda.get_device_properties(dev)
    bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
    kinds = ["bf16", "fp16", "fp16+torch", "fp16+ours"]
    res = {k: [] for k in kinds}
    for _ in range(2):
        for k in kinds:
            res[k].append(run(k))
    for k in kinds:
        print(f"{k:<12} " + "   ".join(f"{v[0]:7.2f} ms/step @ {v[1]} MHz {v[2]} W" for v in res[k]), flush=True)


if __name__ == "__main__":
    main()
