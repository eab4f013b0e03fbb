#!/usr/bin/env python
"""Zero-shot segmentation evaluation with the whole per-image body on the GPU.

Same command line, config, checkpoint and data layout as the reference's tools/seg_evaluation.py; the differences are in
how the work is laid out, not in what is computed:
  * the prompt-ensemble class embeddings are one batched text-tower call (reference: one batch per class, :57-75);
  * images are processed `--batch` at a time and the similarity map is one fused kernel for all classes (reference: one
    image, up to five GEMVs with host syncs, :99-143);
  * candidate selection, min-max normalisation, binarisation, 7x7 dilate/erode, nearest resize, score-weighted argmax and
    the IoU histograms run on the device (simseg_amd.segpost, :112-170), and so does the DenseCRF between normalisation and
    morphology (:31-54 / :153: mean-field inference on two permutohedral lattices, simseg_dense_crf).  `--no-crf` stops at the
    CRF's unary decision; with pydensecrf installed `--host-crf` routes the maps through the library on the host instead
    (cross-check of the device CRF against the package the reference uses).

    python tools/seg_eval_device.py --cfg configs/clip/simseg.vit-b.yaml --ckpt_path ckpts/simseg.vit-b.pth
    python tools/seg_eval_device.py --cfg configs/clip/simseg.vit-s.yaml --synthetic 64        # no data / checkpoint needed
"""
import argparse
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def parse_args():
    ap = argparse.ArgumentParser(description="SimSeg zero-shot segmentation evaluation (device post-processing)")
    ap.add_argument("--cfg", required=True)
    ap.add_argument("--local_rank", "--local-rank", type=int, default=int(os.environ.get("LOCAL_RANK", 0)))      # (torch.distributed.run sets LOCAL_RANK; torch.distributed.launch passes the flag)
    ap.add_argument("--ckpt_path", default="")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--synthetic", type=int, default=0, help="evaluate N synthetic images with random weights")
    ap.add_argument("--no-crf", action="store_true", help="skip the DenseCRF (binary map = its unary decision)")
    ap.add_argument("--host-crf", action="store_true", help="DenseCRF with pydensecrf on the host instead of the device kernels")
    ap.add_argument("--slide", default="", help="WIN,STRIDE: sliding-window evaluation (BASELINE configs[3]): images are fed at their loader size, cut into "
                                                "WIN-pixel windows at STRIDE, per-window similarity maps overlap-averaged (segpost.encode_batch_sliding); "
                                                "transforms.input_size must be WIN")
    return ap.parse_known_args()


def host_crf_refine(images_uint8):
    import numpy as np
    import pydensecrf.densecrf as dcrf

    def refine(prob, cand_idx, cand_score):
        B, K, H, W = prob.shape
        out = torch.zeros(B, K, H, W, dtype=torch.uint8)
        p = prob.cpu().numpy()
        for b in range(B):
            for k in range(K):
                if int(cand_idx[b, k]) < 0:
                    continue
                probs = np.stack([1 - p[b, k], p[b, k]])
                d = dcrf.DenseCRF2D(W, H, 2)
                d.setUnaryEnergy(np.ascontiguousarray(-np.log(probs + 1e-8).reshape(2, -1).astype(np.float32)))
                d.addPairwiseGaussian(sxy=3, compat=3)
                d.addPairwiseBilateral(sxy=40, srgb=13, rgbim=np.ascontiguousarray(images_uint8[b]), compat=10)
                q = np.argmax(np.array(d.inference(3)), axis=0).reshape(H, W)
                out[b, k] = torch.from_numpy((q * 255).astype(np.uint8))
        return out.to(prob.device)
    return refine


def main():
    args, overrides = parse_args()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
    os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    from simseg.core import cfg, init_device, update_cfg
    from simseg.core.hooks.checkpoint import get_dist_state_dict
    from simseg.models import PIPELINE
    from simseg.tasks.clip.config import task_cfg_init_fn, update_clip_config
    from simseg.utils import ENV, build_from_cfg, interpolate_pos_embed, logger
    from simseg.utils.prompt import openai_imagenet_template
    from simseg_amd import ops, segpost
    from simseg_amd.heads import class_text_embeddings, patch_text_similarity

    if args.synthetic:
        overrides = list(overrides) + ["model.image_encoder.pretrained=False", "model.text_encoder.pretrained=False"]
    update_cfg(task_cfg_init_fn, args.cfg, overrides, preprocess_fn=update_clip_config)
    ENV.cfg, ENV.local_rank = cfg, args.local_rank
    init_device(cfg)
    model = build_from_cfg(cfg.model.name, cfg, PIPELINE).to(ENV.device).eval()
    if args.ckpt_path:
        sd = torch.load(args.ckpt_path, map_location="cpu")["state_dict"]
        key = "image_encoder.model.model.pos_embed"
        if key in sd:
            sd[key] = interpolate_pos_embed(sd[key], model.image_encoder.model.model)
        model.load_state_dict(get_dist_state_dict(sd), strict=False)
        logger.emph(f"Loaded ckpt path: {args.ckpt_path}")
    size = cfg.transforms.input_size
    n = size // 16

    def text_matrix(categories):
        if args.synthetic:
            g = torch.Generator().manual_seed(3)
            return torch.nn.functional.normalize(torch.randn(len(categories), 512, generator=g), dim=-1).to(ENV.device)
        from transformers import AutoTokenizer
        tok = AutoTokenizer.from_pretrained(os.environ.get("SIMSEG_TOKENIZER_DIR", cfg.model.text_encoder.tag))
        prompts = [openai_imagenet_template(c) for c in categories]
        enc = tok([t for ts in prompts for t in ts], padding="max_length", truncation=True, max_length=25, return_tensors="pt")
        C, P = len(categories), len(prompts[0])
        with torch.no_grad():
            return class_text_embeddings(model, enc["input_ids"].view(C, P, -1).to(ENV.device), enc["attention_mask"].view(C, P, -1).to(ENV.device))

    def _emit_group(group):
        if all(l.shape == group[0][1].shape and i.shape == group[0][0].shape for i, l in group):
            yield torch.cat([i for i, _ in group]), torch.cat([l for _, l in group])
        else:
            for i, l in group:
                yield i, l

    def batches(name, shard=None):
        """shard = (rank, world): only THIS rank's batches are produced (batch i goes to rank i % world - evaluate_sharded's rule), i.e. a
        rank decodes and transforms 1 / world of the images instead of all of them (the reference's loader gives every rank every image,
        simseg/datasets/seg/seg_dataset.py:67-81).  Batches are formed from `--batch` consecutive dataset items; an item whose label shape
        differs from its batch's ends the batch early in the unsharded form only - the sharded form cuts fixed groups of `--batch` items and
        splits a group with mixed shapes into single-image batches."""
        if args.synthetic:
            g = torch.Generator().manual_seed(1)
            for i, s in enumerate(range(0, args.synthetic, args.batch)):
                b = min(args.batch, args.synthetic - s)
                hw = (size, 2 * size) if args.slide else (size, size)        # sliding window: 1 x 3 windows at half-window stride
                lab = torch.randint(0, 21, (b, *hw), generator=g, dtype=torch.int64).to(torch.uint8)
                img = torch.randn(b, 3, *hw, generator=g)
                if shard is None or i % shard[1] == shard[0]:
                    yield img, lab
            return
        from simseg.datasets.seg.seg_dataset import build_torch_valid_loader
        loader = build_torch_valid_loader(cfg, name, mode="valid")
        if shard is not None and shard[1] > 1:
            # the same loader over a Subset holding this rank's groups of `--batch` consecutive items
            ds = loader.dataset
            idx = [i for i in range(len(ds)) if (i // args.batch) % shard[1] == shard[0]]
            loader = torch.utils.data.DataLoader(torch.utils.data.Subset(ds, idx), batch_size=1, shuffle=False, num_workers=getattr(loader, "num_workers", 0),
                                                 collate_fn=loader.collate_fn)
            group = []
            for image, label in loader:
                group.append((image, label.to(torch.uint8)))
                if len(group) == args.batch:
                    yield from _emit_group(group)
                    group = []
            if group:
                yield from _emit_group(group)
            return
        imgs, labs = [], []
        for image, label in loader:                       # reference loader: batch size 1, labels at raw resolution
            imgs.append(image); labs.append(label.to(torch.uint8))
            same = all(l.shape == labs[0].shape for l in labs)
            if len(imgs) == args.batch or not same:
                keep = len(imgs) if same else len(imgs) - 1
                yield torch.cat(imgs[:keep]), torch.cat(labs[:keep])
                imgs, labs = imgs[keep:], labs[keep:]
        if imgs:
            yield torch.cat(imgs), torch.cat(labs)

    names = ["synthetic"] if args.synthetic else list(cfg.data.valid_name)
    for name in names:
        if args.synthetic:
            cats = [f"class{i}" for i in range(21)]
        else:
            with open(f"data/label_category/{name}.txt") as f:
                cats = [ln.strip() for ln in f]
        top_cls_num = 30 if name == "pascal_context" else 10
        text = text_matrix(cats)
        hist = torch.zeros(3, len(cats), device=ENV.device, dtype=torch.int64)
        mean = torch.tensor(cfg.transforms.normalize.mean, device=ENV.device).view(1, 3, 1, 1)
        std = torch.tensor(cfg.transforms.normalize.std, device=ENV.device).view(1, 3, 1, 1)
        count, t0 = 0, time.perf_counter()
        # two batches in flight on two HIP streams, the next batch's encoder enqueued before the previous batch is finished
        # (segpost.EvalPipeline: the CRF stage's host read and its Python launch loop run under queued MFMA work); the histograms
        # accumulate atomically, so no ordering between batches is needed
        def encode(image, label):
            refine = None
            if args.host_crf and not args.no_crf:
                refine = host_crf_refine((((image * std) + mean) * 255).to(torch.uint8).permute(0, 2, 3, 1).cpu().numpy())
            # (the de-normalised input is what the tool hands to dense_crf, tools/seg_evaluation.py:104)
            st = segpost.encode_batch(model, image, text, top_cls_num, crf=not args.no_crf, mean=mean, std=std, refine=refine)
            st["refine"] = refine
            return st

        def finish(st, image, label):
            return segpost.finish_batch(st, label, hist=hist, refine=st["refine"])

        if args.host_crf and not args.no_crf:
            pipe = segpost.EvalPipeline(ENV.device, encode, finish, pipelined=True)
            with torch.no_grad():
                for image, label in batches(name):
                    image, label = image.to(ENV.device), label.to(ENV.device)
                    pipe.submit(image, label)
                    count += image.shape[0]
                pipe.flush()
            torch.cuda.synchronize()
            iou, miou = segpost.iou_from_hist(hist)
        else:
            # the product loop: batches dealt round-robin to the ranks of the process group (one rank when launched plainly; N under
            # `python -m torch.distributed.run --nproc-per-node N tools/seg_eval_device.py ...`), ONE all-reduce of the [3, C] area histograms
            slide = tuple(int(v) for v in args.slide.split(",")) if args.slide else None
            # (each rank's loader produces only its own batches; evaluate_sharded then sees a one-rank deal of them and still ends with the
            #  all-reduce of the histograms over the world)
            on = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
            shard = (dist.get_rank(), dist.get_world_size()) if on else None
            res = segpost.evaluate_sharded(model, batches(name, shard), text, top_cls_num, slide=slide, crf=not args.no_crf, mean=mean, std=std,
                                           device=ENV.device, presharded=on)
            torch.cuda.synchronize()
            iou, miou, count = res["iou"], res["miou"], res["images"]
        dt = time.perf_counter() - t0
        print(f"---------------- {count} samples evaluated ({name}, {count / dt:.1f} images/s). ----------------")
        logger.emph("multi class iou:", iou)
        logger.emph("final mean iou:", miou)


if __name__ == "__main__":
    main()
