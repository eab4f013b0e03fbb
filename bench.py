#!/usr/bin/env python
"""Benchmark of the SimSeg hot path on MI355X (contract in the task statement / SURVEY.md 8d).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 ...

Workload (BASELINE.json configs[2], weak-scaled): ViT-B/16 + BERT-base contrastive pre-training step on synthetic
CC3M-shaped pairs -- 512 pairs per GPU, 224x224 images, 77-token captions -- forward, global InfoNCE (RCCL embedding
all-gather / reduce-scatter), backward, DDP gradient all-reduce and the AdamW step, all inside the timed region.
Compute dtype bf16 (fp32 accumulate, fp32 master weights / residual stream), matching the reference's autocast recipe.
One step = one pass of the hot path over one batch; value = pairs/s over all ranks.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_BF16 = 2.5e15      # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_F32 = 157.3e12


T_START = time.perf_counter()


def log(msg):
    print(f"[bench +{time.perf_counter() - T_START:7.1f}s] {msg}", file=sys.stderr, flush=True)


TRAFFIC_JSON = "r6_pmc_traffic.json"
_JSON_FD = 1        # where the one JSON line goes (main() parks the real stdout here and points fd 1 at stderr)


def git_blob_sha1(path):
    """git's blob id of a file (sha1 of "blob <size>\\0" + content) without git: the GPU box gets a snapshot with no .git, and the
    PMC traffic entry must be tied to the kernel source it was measured on (`git rev-parse HEAD:simseg_amd/csrc/gemm.hip`)."""
    import hashlib
    try:
        data = open(path, "rb").read()
    except OSError:
        return None
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def flops_per_pair(n_patches, dim, seq, proj=512):
    """Algorithmic FLOPs of one image-text pair, fwd+bwd = 3x fwd."""
    t = n_patches + 1
    f_vit = 12 * (24 * t * dim * dim + 4 * t * t * dim) + 2 * n_patches * 768 * dim
    f_bert = 12 * (24 * seq * 768 * 768 + 4 * seq * seq * 768)
    f_proj = 2 * n_patches * dim * proj + 2 * seq * 768 * proj
    return 3 * (f_vit + f_bert + f_proj)


def build_model(tag, dim, img, rank_argv=()):
    from simseg.core.config import update_cfg
    from simseg.models import PIPELINE
    from simseg.tasks.clip.config import task_cfg_init_fn, update_clip_config
    from simseg.utils import build_from_cfg
    argv = [f"transforms.input_size={img}", f"model.image_encoder.tag={tag}", f"model.image_encoder.embedding_dim={dim}",
            "model.image_encoder.pretrained=False", "model.text_encoder.pretrained=False"] + list(rank_argv)
    cfg = update_cfg(task_cfg_init_fn, os.path.join(REPO, "configs/clip/simseg.vit-b.yaml"), argv, update_clip_config)
    return cfg, build_from_cfg


def synthetic_batch(B, img, L, vocab, seed, device):
    g = torch.Generator().manual_seed(seed)
    image = torch.randn(B, 3, img, img, generator=g)
    lens = torch.randint(8, L + 1, (B,), generator=g)
    ids = torch.randint(1000, vocab, (B, L), generator=g)
    pos = torch.arange(L)[None]
    mask = (pos < lens[:, None]).long()
    ids = ids * mask
    ids[:, 0] = 101
    ids[torch.arange(B), lens - 1] = 102
    # caption_lengths: the captions' token counts as HOST numbers, as a loader has them from its tokenizer before the host->device copy
    # (optional batch key of this package: the text tower then sizes its packed rows without reading the device - simseg_amd/towers.py)
    return {"image": image.to(device), "input_ids": ids.to(device), "attention_mask": mask.to(device), "caption_lengths": lens.clone()}


def _cpu_threads():
    # a bounded thread count: the GPU box reports 256 logical CPUs but oversubscribing them made round-1's first baseline run 100x
    # slower than 8 threads in the build container
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    return torch.get_num_threads()


def cpu_baseline(img, L, budget_s=25.0):
    """The oracle (a torch fp32 restatement of the reference stack) doing the same training step on the host cores: 8 pairs per
    step (large enough to load the cores: 1.2 TFLOP per step)."""
    from oracle import simseg_ref as R
    threads = _cpu_threads()
    B = 8
    ref = R.init_weights_(R.RefCLIP("vit_base_patch16_224_in21k", "bert-base-uncased", img_size=img), seed=2).train()
    opt = torch.optim.AdamW(ref.parameters(), lr=1e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=1e-3)
    b = synthetic_batch(B, img, L, 30522, 7, "cpu")

    def step():
        opt.zero_grad(set_to_none=True)
        loss, _, _ = ref.forward_loss_local(b["image"], b["input_ids"], b["attention_mask"])
        loss.backward()
        opt.step()

    t_warm = time.perf_counter()
    step()
    t_warm = time.perf_counter() - t_warm
    log(f"cpu_baseline: warm-up step of {B} pairs took {t_warm:.1f}s on {threads} threads")
    n, t0 = 0, time.perf_counter()
    while n < 1 or (time.perf_counter() - t0 + t_warm < budget_s and n < 20):
        step()
        n += 1
    dt = time.perf_counter() - t0
    return {"value": round(B * n / dt, 3), "unit": "pairs/s", "cores": threads, "kind": "port",
            "sample": f"{n} fwd+bwd+AdamW steps of {B} pairs (ViT-B/16 @{img}, BERT-base L={L}) with the torch-fp32 oracle"}


def cpu_baseline_seg(img=512, classes=171, windows=1):
    """The seg leg on the host cores, one window: oracle ViT-B tower -> projection -> pooled embedding + similarity map -> candidate
    selection -> min-max map -> DenseCRF (numpy restatement of pydensecrf's inference) -> 7x7 dilate/erode -> resize -> argmax ->
    IoU histograms, i.e. the reference's per-image loop (tools/seg_evaluation.py:99-170) with the oracle's pieces."""
    import numpy as np
    from oracle import crf_ref as CR
    from oracle import segpost_ref as SR
    from oracle import simseg_ref as R
    threads = _cpu_threads()
    n = img // 16
    g = torch.Generator().manual_seed(3)
    vit = R.init_weights_(R.RefViT("vit_base_patch16_224_in21k", img), seed=4).eval()
    proj = torch.randn(512, 768, generator=g) * 0.02
    text = torch.nn.functional.normalize(torch.randn(classes, 512, generator=g), dim=-1)
    image = torch.randn(windows, 3, img, img, generator=g)
    rgb = torch.randint(0, 256, (windows, img, img, 3), generator=g, dtype=torch.int64).numpy().astype(np.uint8)
    labels = torch.randint(0, classes, (windows, img, img), generator=g)
    t0 = time.perf_counter()
    visited = 0
    with torch.no_grad():
        feats = vit(image)[:, 1:]
        tok = feats @ proj.T
        pooled = R.l2norm(R.topk_pool(tok, 5))
        sim = R.seg_similarity(tok, text)
        scores = pooled @ text.T
    for b in range(windows):
        idx, sc, thr = SR.select_candidates(scores[b], 10)
        temp = np.zeros((classes, img, img))
        for k, c in enumerate(idx):
            if c < 0:
                continue
            visited += 1
            norm, _ = SR.normalised_map(sim[b, :, c].numpy(), n)
            m = (CR.dense_crf(rgb[b], norm) * 255).astype(np.uint8)
            m = SR.morph7_fast(SR.morph7_fast(m, False), True)
            temp[c] = m.astype(np.float64) * sc[k]
        SR.intersect_and_union(torch.from_numpy(temp.argmax(0)), labels[b], classes)
    dt = time.perf_counter() - t0
    return {"value": round(windows / dt, 4), "unit": "windows/s", "cores": threads, "kind": "port",
            "sample": f"{windows} window(s) of {img}x{img}, {classes} classes, {visited} candidate map(s) through the DenseCRF restatement (oracle ViT-B "
                      f"tower + per-image loop of tools/seg_evaluation.py)"}


def cpu_baseline_retrieval(m=1000, n=5000, d=512):
    """The retrieval leg on the host cores at reduced size: the reference's argsort + gather + first-match path
    (tasks/clip/hooks/utils.py:35-75) as restated in the oracle, both directions."""
    from oracle import simseg_ref as R
    threads = _cpu_threads()
    g = torch.Generator().manual_seed(5)
    img = torch.nn.functional.normalize(torch.randn(m, d, generator=g), dim=-1)
    txt = torch.nn.functional.normalize(img.repeat_interleave(n // m, 0) + 0.08 * torch.randn(n, d, generator=g), dim=-1)
    gi, gt = torch.arange(m), torch.arange(n) // (n // m)
    R.retrieval_recalls(img, gi, txt, gt)
    reps, t0 = 0, time.perf_counter()
    while reps < 1 or time.perf_counter() - t0 < 5.0:
        R.retrieval_recalls(img, gi, txt, gt)
        R.retrieval_recalls(txt, gt, img, gi)
        reps += 1
    dt = (time.perf_counter() - t0) / reps
    return {"value": round(2.0 * m * n / dt, 1), "unit": "similarities/s", "cores": threads, "kind": "port",
            "sample": f"{reps} evaluations of {m}x{n}x{d}, both directions (argsort + gather + first match, the reference's path)"}


def seg_eval_bench(dev, world, dtype, windows=63, steps=3, img=512, classes=171, tag="vit_base_patch16_224_in21k", dim=768, crf=False):
    """Zero-shot segmentation GPU stage (BASELINE configs[3] shape): ViT-B on 512x512 windows -> projection -> LoDA pooled
    embedding + dense patch x class-text similarity map for all `classes` + candidate selection, masks, 7x7 morphology,
    resize/argmax and IoU histograms on the device (tools/seg_evaluation.py:99-170 without the CPU CRF stage).  Independent
    windows: sharded over ranks, no collective on the data path (one all-reduce of the [3,C] histograms at the end).
    Batch = 21 source images of 3 windows: 63 x 1025 tokens = 252.2 row panels of 256, i.e. the GEMM tile grids (759 / 2277 /
    3036 tiles) fill their last round of 256 CUs to 97-99 %; 64 windows would be 257 panels = 3.01 rounds for N = 768.  The bf16 legs run 256
    windows per batch instead (round 3): every GEMM row count is then a multiple of 256 - full tiles only - and the persistent ping-pong
    kernel takes them (+5 % ViT-B, +15 % ViT-S; the exact-mode legs measured no better with it: tools/seg_windows_ab.py)."""
    from simseg_amd.heads import patch_text_similarity
    from simseg_amd import ops
    from simseg.models import PIPELINE
    os.environ["SIMSEG_AMD_COMPUTE"] = dtype
    cfg, build = build_model(tag, dim, img)
    torch.manual_seed(7)
    model = build(cfg.model.name, cfg, PIPELINE).to(dev).eval()
    g = torch.Generator().manual_seed(3)
    text = torch.nn.functional.normalize(torch.randn(classes, 512, generator=g), dim=-1).to(dev)
    images = torch.randn(windows, 3, img, img, generator=g).to(dev)
    labels = torch.randint(0, classes, (windows, img, img), generator=g, dtype=torch.int64).to(torch.uint8)
    labels[torch.rand(windows, img, img, generator=g) < 0.05] = 255
    labels = labels.to(dev)
    # de-normalised network inputs for the DenseCRF (tools/seg_evaluation.py:104): smooth colour fields + noise, so that the
    # bilateral kernel has structure to follow
    yy, xx = torch.meshgrid(torch.arange(img), torch.arange(img), indexing="ij")
    base_rgb = torch.stack([(xx * 255 // img), (yy * 255 // img), ((xx + yy) * 127 // img)], -1).float()
    images_u8 = (base_rgb[None] + 20 * torch.randn(min(windows, 8), img, img, 3, generator=g)).clamp(0, 255).to(torch.uint8)
    images_u8 = images_u8.repeat((windows + 7) // 8, 1, 1, 1)[:windows].contiguous().to(dev)
    hist = torch.zeros(3, classes, device=dev, dtype=torch.int64)
    cdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    from simseg_amd import segpost
    post_ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]

    def encode(ev=None):
        with torch.no_grad():
            feats = model.forward_image_feature(images)                 # [B, 1024, 768]
            pooled = model.forward_image_project(feats)                 # [B, 512]
            tok = model.image_projection(feats)                         # [B, 1024, 512]
            sim = patch_text_similarity(tok, text, compute_dtype=cdt)   # [B, 1024, classes]
            scores = ops.gemm(pooled, text)                             # [B, classes]
            # post-processing of tools/seg_evaluation.py:112-170 on the device, first half: candidate classes + min-max maps
            if ev is not None:
                ev.record()
            st = segpost.segment_begin(sim, scores, img // 16, 10, need_prob=crf)
        return sim, scores, st

    def finish(enc):
        with torch.no_grad():
            out = segpost.segment_finish(enc[2], labels, hist=hist, want_pred=False, images_u8=images_u8 if crf else None)
        return enc[0], enc[1], out

    def step():                      # one batch alone (the post-processing stage timed by itself)
        enc = encode(post_ev[0])
        out = finish(enc)
        post_ev[1].record()
        return out

    # Two batches in flight on two HIP streams, software-pipelined on the host as tools/seg_eval_device.py does (segpost.EvalPipeline):
    # batch i's encoder is enqueued BEFORE batch i-1 is finished, so the DenseCRF stage's host read (the candidate table) finds its data
    # long finished and the GPU has MFMA work queued while Python walks the stage's launch loop; one batch's attention / LayerNorm /
    # post-processing phases fill the tile-grid tails of the other's GEMMs.  The histograms accumulate atomically.
    def run(n_batches):
        pipe = segpost.EvalPipeline(dev, lambda: encode(), lambda enc: finish(enc), pipelined=crf and os.environ.get("SIMSEG_SEG_PIPELINE", "1") != "0")
        for _ in range(n_batches):
            pipe.submit()
        return pipe.flush()

    step()
    torch.cuda.synchronize()
    step()                           # post-processing alone: events around the second half of one batch, nothing else on the GPU
    torch.cuda.synchronize()
    post_ms = post_ev[0].elapsed_time(post_ev[1])
    run(2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = run(2 * steps)
    torch.cuda.synchronize()
    el = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    steps = 2 * steps                                    # batches processed
    visited = int((last[2]["cand_idx"] >= 0).sum())
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(hist)               # the only exchange of the seg path - intersect / area histograms, once
    del model
    n_patches = (img // 16) ** 2
    t = n_patches + 1
    fl = 12 * (24 * t * dim * dim + 4 * t * t * dim) + 2 * n_patches * 768 * dim + 2 * 2 * n_patches * dim * 512 + 2 * n_patches * 512 * classes
    wps = world * windows * steps / float(el)
    # algorithmic bytes of the post stage: every VISITED candidate map (img x img bytes) is written once, read and written by the
    # fused dilate+erode, and read again for the argmax / IoU pass; the label map is read once per window.  (All five slots are
    # read by the argmax pass as the reference's temp_pred[...] stack would be: counted for visited ones only.)
    post_bytes = img * img * (4 * visited + windows)
    out = {"post_ms_per_step": round(post_ms, 3), "post_visited_candidates_per_window": round(visited / windows, 2),
           "windows_per_s": round(wps, 1), "dtype": dtype, "window": img,
           "classes": classes, "windows_per_batch": windows, "batches_in_flight": 2, "dense_crf": bool(crf),
           "tflops_per_gpu": round(wps / world * fl / 1e12, 1)}
    if crf:
        # The DenseCRF stage is hash-table / scattered-gather work; its traffic is not derivable from tensor sizes.  What it moves per
        # window comes from the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/prof_crf.sh (profiles/r4_crf_traffic.json, when
        # that file was measured on this tree's crf.hip) - never from dividing the similarity-map bytes by the stage time.
        out["post_stage"] = "candidate maps -> DenseCRF (permutohedral mean field) -> 7x7 closing -> resize + argmax + histograms"
        try:
            with open(os.path.join(REPO, "profiles", "r4_crf_traffic.json")) as f:
                tj = json.load(f)
            ent = tj.get(f"{img}")
            if ent and tj.get("crf_hip_blob") == git_blob_sha1(os.path.join(REPO, "simseg_amd", "csrc", "crf.hip")):
                mb = ent["hbm_mb_per_window"]
                out["post_hbm_MB_per_window_pmc"] = mb
                out["post_GBps_pmc"] = round(mb * windows / post_ms, 1)               # MB per ms = GB/s
                out["post_frac_of_hbm_peak_pmc"] = round(mb * windows / post_ms / 8000.0, 4)
        except (OSError, ValueError, KeyError):
            pass
    else:
        out["post_GBps"] = round(post_bytes / post_ms / 1e6, 1)
        out["post_frac_of_hbm_peak"] = round(post_bytes / post_ms / 1e6 / 8000.0, 4)
    if dtype == "bf16":
        out["frac_of_peak"] = round(wps / world * fl / PEAK_BF16, 4)
    else:
        # Exact mode runs on the bf16 matrix pipe since round 3 (six bf16 piece products per fp32 product: simseg_split_bf16x3,
        # attn_fwd_x3), so the hardware fraction is 6 x the algorithmic fp32 FLOPs against the bf16 peak; the fraction of the fp32 MFMA
        # peak is what an fp32-MFMA implementation would have to reach for the same throughput (it can exceed 1 for that reason).
        out["frac_of_peak"] = round(6.0 * wps / world * fl / PEAK_BF16, 4)
        out["frac_of_peak_counts"] = "6 bf16 piece products per algorithmic fp32 product, against the dense bf16 MFMA peak (the pipe the kernels use)"
        out["fp32_equivalent_frac_of_fp32_mfma_peak"] = round(wps / world * fl / PEAK_F32, 4)
    return out


def seg_slide_bench(dev, world, dtype, images=21, steps=3, H=512, W=1024, win=512, stride=256, classes=171, tag="vit_base_patch16_224_in21k",
                    dim=768, crf=True, window_batch=None):
    """BASELINE configs[3] as SURVEY.md 8d cfg 4 defines it: COCO-Stuff-shaped 512 x 1024 source images, 3 windows of 512^2 at stride 256
    each, ViT-B towers per window, per-window patch x class-text similarity maps OVERLAP-AVERAGED on the source image's 32 x 64 patch grid
    (simseg_stitch_windows), then ONE per-image body (candidates, min-max, DenseCRF on the 512 x 1024 image, closing, argmax, IoU areas) per
    source image; images/s counts SOURCE images.  Source images are independent: every rank runs its own batches (weak shard) and the
    [3, C] histograms meet in one all-reduce at the end (segpost.evaluate_sharded is the product form of this loop)."""
    from simseg.models import PIPELINE
    from simseg_amd import segpost
    os.environ["SIMSEG_AMD_COMPUTE"] = dtype
    cfg, build = build_model(tag, dim, win)
    torch.manual_seed(7)
    model = build(cfg.model.name, cfg, PIPELINE).to(dev).eval()
    g = torch.Generator().manual_seed(3)
    text = torch.nn.functional.normalize(torch.randn(classes, 512, generator=g), dim=-1).to(dev)
    # network inputs that ARE images (the DenseCRF reads them back de-normalised): smooth colour fields + noise, normalised as the transforms do
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    base = torch.stack([(xx * 255 // W), (yy * 255 // H), ((xx + yy) * 127 // max(H, W))], 0).float()
    nb = min(images, 4)
    rgb = (base[None] + 20 * torch.randn(nb, 3, H, W, generator=g)).clamp(0, 255)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    x = ((rgb / 255.0 - mean) / std).repeat((images + nb - 1) // nb, 1, 1, 1)[:images].contiguous().to(dev)
    labels = torch.randint(0, classes, (images, H, W), generator=g, dtype=torch.int64).to(torch.uint8)
    labels[torch.rand(images, H, W, generator=g) < 0.05] = 255
    labels = labels.to(dev)
    mean, std = mean.to(dev), std.to(dev)
    hist = torch.zeros(3, classes, device=dev, dtype=torch.int64)
    cdt = torch.bfloat16 if dtype == "bf16" else None
    wy, wx = segpost.window_grid(H, W, win, stride)

    def encode():
        with torch.no_grad():
            return segpost.encode_batch_sliding(model, x, text, 10, win=win, stride=stride, crf=crf, mean=mean, std=std, sim_dtype=cdt, window_batch=window_batch)

    def finish(st):
        with torch.no_grad():
            return segpost.finish_batch(st, labels, hist=hist)

    def run(n_batches):
        pipe = segpost.EvalPipeline(dev, encode, finish, pipelined=crf and os.environ.get("SIMSEG_SEG_PIPELINE", "1") != "0")
        for _ in range(n_batches):
            pipe.submit()
        return pipe.flush()

    run(2)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    last = run(2 * steps)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(hist, op=dist.ReduceOp.SUM)
    visited = int((last["cand_idx"] >= 0).sum())
    ips = world * images * 2 * steps / float(el)
    del model
    torch.cuda.empty_cache()
    return {"images_per_s": round(ips, 1), "windows_per_s": round(ips * wy * wx, 1), "source_image": [H, W], "window": win, "stride": stride,
            "windows_per_image": wy * wx, "stitched_patch_grid": [H // 16, W // 16], "images_per_batch": images, "classes": classes, "dtype": dtype,
            "dense_crf": bool(crf), "batches_in_flight": 2, "visited_candidates_per_image": round(visited / images, 2),
            "pixels_labelled": int(hist[2].sum()), "stitch": "overlap-average of the per-window [32,32,C] maps on the [32,64] grid (simseg_stitch_windows), image scores = mean of window scores"}


def seg_latency_bench(dev, dtype, img, classes, tag, dim, reps=50):
    """One image per call, as the reference's tool runs (tools/seg_evaluation.py:99-170, batch size 1): towers -> projection ->
    similarity map -> device post-processing.  At this size the forward is ~150 launches of a few microseconds, so the same
    pipeline is also captured once into a hipGraph (simseg_amd/graph.py) and replayed.  Returns ms per image, both ways."""
    from simseg.models import PIPELINE
    from simseg_amd import ops, segpost
    from simseg_amd.graph import GraphedCall
    from simseg_amd.heads import patch_text_similarity
    os.environ["SIMSEG_AMD_COMPUTE"] = dtype
    cfg, build = build_model(tag, dim, img)
    torch.manual_seed(7)
    model = build(cfg.model.name, cfg, PIPELINE).to(dev).eval()
    g = torch.Generator().manual_seed(3)
    text = torch.nn.functional.normalize(torch.randn(classes, 512, generator=g), dim=-1).to(dev)
    image = torch.randn(1, 3, img, img, generator=g).to(dev)
    labels = torch.randint(0, classes, (1, img, img), generator=g, dtype=torch.int64).to(torch.uint8).to(dev)
    cdt = torch.bfloat16 if dtype == "bf16" else torch.float32

    def pipeline(im):
        feats = model.forward_image_feature(im)
        pooled = model.forward_image_project(feats)
        sim = patch_text_similarity(model.image_projection(feats), text, compute_dtype=cdt)
        out = segpost.segment(sim, ops.gemm(pooled.float(), text), labels, img // 16, 10, want_pred=True)
        return out["pred"], out["hist"]

    def timed(fn):
        for _ in range(5):
            fn(image)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn(image)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    with torch.no_grad():
        eager = timed(pipeline)
    graphed = timed(GraphedCall(pipeline, image))
    del model
    return {"dtype": dtype, "encoder": tag, "input": img, "classes": classes, "eager_ms_per_image": round(eager, 3),
            "hipgraph_ms_per_image": round(graphed, 3), "images_per_s_hipgraph": round(1e3 / graphed, 1)}


def retrieval_bench(dev, m=5000, n=25000, d=512, reps=5):
    """BASELINE configs[4] shape: R@1/5/10 in both directions over the full 5k x 25k similarity matrix (fp32 MFMA GEMM +
    first-match-rank kernel instead of the reference's argsort + int64 gid gather, hooks/utils.py:36-42)."""
    from simseg_amd.heads import retrieval_recalls_both
    g = torch.Generator().manual_seed(5)
    img = torch.nn.functional.normalize(torch.randn(m, d, generator=g), dim=-1).to(dev)
    txt = torch.nn.functional.normalize(img.repeat_interleave(n // m, 0) + 0.08 * torch.randn(n, d, generator=g).to(dev), dim=-1)
    gi, gt = torch.arange(m, device=dev), torch.arange(n, device=dev) // (n // m)
    retrieval_recalls_both(img, gi, txt, gt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        a, b = retrieval_recalls_both(img, gi, txt, gt)      # ONE similarity matrix: rows rank columns (i2t), columns rank rows (t2i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    # "similarities" counts both directions' M x N scores as the reference's two calls produce them (2 M N), computed here once
    return {"shape": f"{m}x{n}x{d}, both directions", "ms_per_eval": round(dt * 1e3, 3), "similarities_per_s": round(2.0 * m * n / dt, 1),
            "gemm_tflops_fp32_equivalent": round(2.0 * m * n * d / dt / 1e12, 1),
            "gemm_note": "the whole evaluation's time (GEMM + two rank passes) divided into the algorithmic fp32 FLOPs; the matrix itself is the "
                         "split-bf16 form: 6 bf16 piece products per fp32 product on the bf16 MFMA pipe",
            "gemm_launches_per_eval": 1,
            "i2t_R@1": round(a["R@1"], 4), "t2i_R@1": round(b["R@1"], 4)}


def retrieval_multi_rank_bench(dev, rank, world, n_img=5000, cap=5, d=512, reps=3):
    """The retrieval evaluation as N ranks run it (tools/retrieval_evaluation.py:65-99): every rank holds the embeddings of its shard
    of the 25000 (image, caption) rows, the shards are all-gathered (equal counts: short shards padded with image_id = -1, dropped
    after the gather), rank 0 computes both recall directions.  Timed: gather + metric (the encoders are timed by encoder_inclusive)."""
    from simseg_amd.retrieval import evaluate_sharded
    g = torch.Generator().manual_seed(5)
    img = torch.nn.functional.normalize(torch.randn(n_img, d, generator=g), dim=-1)
    rows = n_img * cap
    txt = torch.nn.functional.normalize(img.repeat_interleave(cap, 0) + 0.08 * torch.randn(rows, d, generator=g), dim=-1)
    per = (rows + world - 1) // world + (7 if rank == 0 else 0)          # uneven on purpose: the padding path runs
    lo = min(rows, rank * ((rows + world - 1) // world) + (0 if rank == 0 else 7))
    hi = min(rows, lo + per)
    if rank == world - 1:
        hi = rows
    shard = {"image_embeddings": img.repeat_interleave(cap, 0)[lo:hi].contiguous().to(dev), "text_embeddings": txt[lo:hi].contiguous().to(dev),
             "image_id": (torch.arange(rows) // cap)[lo:hi].to(dev), "caption_id": torch.arange(rows)[lo:hi].to(dev)}
    out = evaluate_sharded(shard)
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = evaluate_sharded(shard)
    dist.barrier()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    if rank != 0:
        return None
    return {"ranks": world, "rows": rows, "ms_per_eval_incl_gather": round(dt * 1e3, 3), "I2T_R@1": round(out["coco_I2T-R@1"], 3),
            "T2I_R@1": round(out["coco_T2I-R@1"], 3), "RSUM": round(out["coco_RSUM"], 3)}


def retrieval_encode_bench(dev, n_img=5000, cap_per_img=5, img=288, L=25, ib=250):
    """Encoder-inclusive retrieval eval (README.md:185 / tools/retrieval_evaluation.py:60-100 at MSCOCO-5k shape): 5000 images at
    288^2 and 25000 captions of 25 tokens through the towers (bf16), then recalls in both directions.  Synthetic inputs,
    random weights; returns wall time and rates."""
    from simseg.models import PIPELINE
    from simseg_amd.heads import retrieval_recalls_both
    os.environ["SIMSEG_AMD_COMPUTE"] = "bf16"
    cfg, build = build_model("vit_base_patch16_224_in21k", 768, img)
    torch.manual_seed(11)
    model = build(cfg.model.name, cfg, PIPELINE).to(dev).eval()
    tb = ib * cap_per_img
    batch = synthetic_batch(tb, img, L, 30522, 77, dev)
    batch["image"] = batch["image"][:ib].contiguous()

    def run():
        ie, te = [], []
        with torch.no_grad():
            for _ in range(n_img // ib):
                a, b = model(batch, embeddings="all")
                ie.append(a); te.append(b)
            ie, te = torch.cat(ie).float(), torch.cat(te).float()
            gi = torch.arange(ie.shape[0], device=dev)
            gt = torch.arange(te.shape[0], device=dev) // cap_per_img
            return retrieval_recalls_both(ie, gi, te, gt)

    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    del model
    return {"shape": f"{n_img} images @{img}^2 + {n_img * cap_per_img} captions L={L}, bf16 towers", "seconds": round(dt, 3),
            "images_per_s": round(n_img / dt, 1), "captions_per_s": round(n_img * cap_per_img / dt, 1)}


def attention_roofline(dev, B, img, L, H=12, reps=10, dtype=torch.bfloat16):
    """The 16-bit attention kernels of the step, each timed alone (events, `reps` launches) on the step's own shapes: the image tower's
    (B, T = 1 + (img/16)^2) and the text tower's (B, L) with dropout and a ragged key-padding mask.  FLOPs = 4 T^2 64 per head forward,
    2.5x that backward (five tile products against two); bytes = qkv read + ctx written (forward), + dO, O read and dqkv written (backward).
    Both fractions are given: at these lengths the kernels are bound by HBM traffic and softmax VALU work, not by the matrix pipe."""
    from simseg_amd import ops
    out = {}
    g = torch.Generator(device=dev).manual_seed(0)

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3

    for name, T, drop in (("attention_image_tower", 1 + (img // 16) ** 2, 0.0), ("attention_text_tower", L, 0.1)):
        qkv = torch.randn(B, T, 3 * H * 64, device=dev, generator=g).to(dtype)
        mask, lens2 = None, float(T) * T * B
        if drop:
            lens = torch.randint(8, T + 1, (B,), device=dev, generator=g)
            mask = (torch.arange(T, device=dev)[None] < lens[:, None]).long()
            lens2 = float((lens.double() ** 2).sum())
        out_, lse = ops.attention_fwd(qkv, H, mask, scale=0.125, save_lse=True, drop_seed=3, drop_p=drop)
        do = torch.randn_like(out_)
        tf = timed(lambda: ops.attention_fwd(qkv, H, mask, scale=0.125, save_lse=True, drop_seed=3, drop_p=drop))
        tb = timed(lambda: ops.attention_bwd(qkv, out_, do, lse, H, mask, scale=0.125, drop_seed=3, drop_p=drop))
        fl = 4.0 * 64 * H * (lens2 if mask is not None else float(T) * T * B)
        by = 2.0 * B * T * H * 64 * 4                     # q, k, v read + ctx written, 2 bytes each
        for tag, sec, f, b in (("fwd", tf, fl, by), ("bwd", tb, 2.5 * fl, by * 2.0)):
            out[f"{name}_{tag}"] = {"shape": f"B={B} T={T} H={H}" + (" ragged mask, dropout 0.1" if drop else ""), "ms": round(sec * 1e3, 4),
                                    "tflops": round(f / sec / 1e12, 1), "frac": round(f / sec / PEAK_BF16, 4),
                                    "GBps": round(b / sec / 1e9, 1), "frac_of_hbm_peak": round(b / sec / 8e12, 4)}
        del qkv, out_, lse, do
    return out


def layernorm_roofline(dev, B, img, D=768, reps=10, dtype=torch.bfloat16):
    """The LayerNorm kernels of the ViT blocks, each timed alone on the step's shape ([B * T, D]), in the forms the 16-bit step runs: forward
    fp32 -> 16-bit (6 bytes per element) and backward dy16 + y16 + dres16 -> dx16 (8 bytes per element, ln_bwd16_kernel + its column reduce)."""
    from simseg_amd import ops
    T = 1 + (img // 16) ** 2
    M = B * T
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(M, D, device=dev, generator=g)
    w, b = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    y, _, mean, rstd = ops.layernorm_fwd(x, w, b, 1e-6, out_dtype=dtype, save_stats=True)
    dy, dres = torch.randn(M, D, device=dev, generator=g).to(dtype), torch.randn(M, D, device=dev, generator=g).to(dtype)
    dg, db, ds = torch.zeros(D, device=dev), torch.zeros(D, device=dev), torch.zeros(D, device=dev)

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3

    tf = timed(lambda: ops.layernorm_fwd(x, w, b, 1e-6, out_dtype=dtype, save_stats=True))
    tb = timed(lambda: ops.layernorm_bwd(x, mean, rstd, w, dg, db, dy16=dy, dres16=dres, dxsum=ds, want_f32=False, y16=y, beta=b))
    out = {}
    for tag, sec, by in (("fwd", tf, 6.0 * M * D), ("bwd", tb, 8.0 * M * D)):
        out[f"layernorm_vit_{tag}"] = {"shape": f"[{M}, {D}]", "ms": round(sec * 1e3, 4), "bound": "hbm", "GBps": round(by / sec / 1e9, 1),
                                       "frac_of_hbm_peak": round(by / sec / 8e12, 4), "launches_per_step": 25 if tag == "fwd" else 24}
    return out


class ClockSampler:
    """Shader clock and package power during the timed region (rocm-smi polled from a side thread; host-side only).  MI355X is
    power-capped under matrix-core load: the dense peaks of MI355X_MICROARCH.md assume 2.4 GHz, the step runs at ~1.9 GHz / ~1.36 kW,
    a long-K GEMM alone at ~1.7 GHz / 1.4 kW (profiles/r2_clock_power_under_load.txt)."""

    def __init__(self, period=0.3, bdf=None):
        import threading
        self.samples, self._stop, self.period, self.t_from, self.bdf = [], False, period, 0.0, bdf
        self.th = threading.Thread(target=self._run, daemon=True)
        self.th.start()

    def _sysfs(self):
        """(sclk MHz, package W) straight from the amdgpu sysfs files rocm-smi reads - no fork / exec of a Python tool from a process
        that has tens of GB mapped while the timed steps are being launched.  The card is the one whose PCI address is this process's
        HIP device (a node shows all eight under /sys); None when that is unknown or the files are not there."""
        import glob
        import re
        if not self.bdf:
            return None
        for dev in sorted(glob.glob("/sys/class/drm/card*/device")):
            try:
                if not os.path.realpath(dev).lower().endswith(self.bdf):
                    continue
                with open(dev + "/pp_dpm_sclk") as f:
                    cur = [ln for ln in f.read().splitlines() if ln.strip().endswith("*")]
                mhz = int(re.search(r"(\d+)\s*[Mm][Hh]z", cur[0]).group(1))
                for name in ("power1_average", "power1_input"):
                    hw = glob.glob(dev + "/hwmon/hwmon*/" + name)
                    if hw:
                        with open(hw[0]) as f:
                            return mhz, int(f.read().strip()) / 1e6
            except Exception:       # noqa: BLE001
                continue
        return None

    @staticmethod
    def _smi():
        import re
        import subprocess
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
        m = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
        w = re.search(r"Power \(W\): ([0-9.]+)", out)
        return (int(m.group(1)), float(w.group(1))) if m and w else None

    def _run(self):
        read = self._sysfs if self._sysfs() is not None else self._smi
        self.source = "amdgpu sysfs" if read == self._sysfs else "rocm-smi"
        while not self._stop:
            try:
                v = read()
                if v:
                    self.samples.append((time.perf_counter(), v[0], v[1]))
            except Exception:       # noqa: BLE001  (no rocm-smi either: the fields stay null)
                return
            time.sleep(self.period)

    def mark(self):
        """Samples from here on count (the sampler is started BEFORE the warm-up steps, so that its own start-up - a cold rocm-smi on
        a fresh box, if the sysfs files are missing - does not fall into the timed region)."""
        self.t_from = time.perf_counter()

    def stop(self):
        self._stop = True
        self.th.join(timeout=15)
        s = [x[1:] for x in self.samples if x[0] >= self.t_from]
        if not s:
            return None
        sclk = sum(x[0] for x in s) / len(s)
        return {"sclk_mhz_avg": round(sclk, 0), "power_w_avg": round(sum(x[1] for x in s) / len(s), 0), "samples": len(s),
                "nominal_sclk_mhz": 2400, "dense_bf16_peak_at_this_clock_tflops": round(PEAK_BF16 / 1e12 * sclk / 2400.0, 0),
                "source": getattr(self, "source", None)}


def guarded(name, fn, *a, all_ranks=True, **kw):
    """A secondary leg must not take the headline down with it: its exception becomes {"error": ...} in the line (and a log entry).  With
    N > 1 ranks the verdict is agreed on after the leg (one all-reduce of an ok flag): a leg that failed on ANY rank is reported as failed by
    every rank, so no rank carries on with results the others do not have.  (A rank that dies INSIDE a leg's collective still leaves the
    others waiting there until the process group's timeout - the legs' device work is the same on every rank, which makes that unlikely.)"""
    res, err = None, None
    try:
        res = fn(*a, **kw)
    except Exception as e:       # noqa: BLE001
        import traceback
        log(f"leg {name} FAILED: {e!r}\n{traceback.format_exc()}")
        try:
            torch.cuda.synchronize()
        except Exception:       # noqa: BLE001
            pass
        err = {"error": f"{type(e).__name__}: {e}"[:500]}
    if all_ranks and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:      # (all_ranks=False: a leg only rank 0 runs)
        try:
            bad = torch.tensor([0 if err is None else 1], device="cuda" if dist.get_backend() == "nccl" else "cpu", dtype=torch.int32)
            dist.all_reduce(bad, op=dist.ReduceOp.MAX)
            if int(bad) and err is None:
                err = {"error": f"leg {name} failed on another rank"}
        except Exception as e:       # noqa: BLE001
            err = err or {"error": f"leg {name}: ranks could not agree on its outcome ({type(e).__name__})"}
    return res if err is None else err


def self_launch(n):
    """`python bench.py --gpus N` outside a launcher: become `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same
    arguments>` - one rank per GPU over RCCL, rendezvous on 127.0.0.1 and a free port (the reference: launch.py:33-70 builds
    `torch.distributed.launch --nproc_per_node`, simseg/core/initial.py:54 calls init_process_group from the environment).  The ranks inherit
    stdout, so rank 0's one JSON line is still the only thing on it; the launcher's exit code is this process's."""
    import socket
    try:
        have = torch.cuda.device_count()
    except Exception:       # noqa: BLE001
        have = 0
    if have < n and "SIMSEG_BENCH_DEVICE" not in os.environ:
        raise SystemExit(f"bench.py: --gpus {n} but {have} HIP device(s) are visible")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL's intra-node transport fails with the legacy mode on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("self-launch: " + " ".join(cmd))
    sys.stdout.flush(); sys.stderr.flush()
    os.execvpe(cmd[0], cmd, env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs-per-gpu", type=int, default=512)
    ap.add_argument("--img", type=int, default=224)
    ap.add_argument("--seq-len", type=int, default=77)
    ap.add_argument("--tag", default="vit_base_patch16_224_in21k")
    ap.add_argument("--batches", type=int, default=4, help="distinct synthetic batches rotated through the steps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-seg", action="store_true", help="skip the zero-shot-seg eval stage measurement")
    ap.add_argument("--retrieval", action="store_true", help="with --no-seg: still run the (quick) retrieval metric legs")
    ap.add_argument("--local_rank", "--local-rank", type=int, default=None,
                    help="set by torch.distributed.launch-style launchers (the reference's launch.py:33-70); torch.distributed.run sets LOCAL_RANK instead")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)                             # does not return: this process becomes the launcher of N ranks
    # stdout carries the ONE JSON line and nothing else: file descriptor 1 is parked and re-pointed at stderr for the whole run, so whatever a
    # library writes to the C-level stdout (RCCL prints a version banner there, flushed at exit - i.e. AFTER the line - when stdout is a
    # pipe) cannot land next to it; rank 0 writes the line to the parked descriptor at the end.
    global _JSON_FD
    sys.stdout.flush()
    _JSON_FD = os.dup(1)
    os.dup2(2, 1)
    HEAD = os.environ.get("SIMSEG_BENCH_HEADLINE", "fp16").lower()      # the headline step's arithmetic: the reference's own AMP type (fp16 +
    assert HEAD in ("fp16", "bf16"), HEAD                               #  a live GradScaler, clip_runner.py:226-230); "bf16": round 1-4's headline
    os.environ["SIMSEG_AMD_COMPUTE"] = HEAD
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("SIMSEG_BENCH_DEVICE", os.environ.get("LOCAL_RANK", args.local_rank if args.local_rank is not None else 0)))     # override: bring-up of N ranks on one GPU
    FULL = world == 1 or os.environ.get("SIMSEG_BENCH_FULL", "0") == "1"       # N > 1: the headline, the other 16-bit type and the sharded evaluation legs only
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (launch with `python bench.py --gpus N`, which starts the N ranks "
                         f"itself, or under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`)")
    from simseg.utils import ENV, logger
    logger.STREAM = sys.stderr          # stdout carries the one JSON line only
    ENV.local_rank = local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    cpu = cpu_seg = cpu_retr = None       # (the CPU legs run AFTER every GPU leg, below: 30-60 s of 32-thread host work in front of the timed
                                          #  steps once left a box's host side slow enough to starve the GPU - 93.5 instead of 87.6 ms per step)

    log("cpu baseline done" if cpu else "no cpu baseline")
    from simseg.core import init_device
    from simseg_amd import ops
    from simseg_amd.nn import VIT_ARCH
    from simseg_amd.optim import AdamW
    dim = VIT_ARCH[args.tag]["dim"]
    cfg, build = build_model(args.tag, dim, args.img)
    init_device(cfg)                                   # RCCL process group (env://), ENV.rank/size/device
    # what the process group itself reports (not the environment): backend, world size and the device every rank computes on
    pg_info = {"backend": dist.get_backend() if dist.is_initialized() else None,
               "world_size": dist.get_world_size() if dist.is_initialized() else 1}
    me = {"rank": rank, "device": f"cuda:{local}", "name": torch.cuda.get_device_name(dev)}
    try:
        pr0 = torch.cuda.get_device_properties(dev)
        me["pci"] = f"{pr0.pci_domain_id:04x}:{pr0.pci_bus_id:02x}:{pr0.pci_device_id:02x}.0"
    except Exception:       # noqa: BLE001
        pass
    if world > 1:
        ranks = [None] * world
        dist.all_gather_object(ranks, me)
        pg_info["ranks"] = ranks
        if len({r.get("pci", r["device"]) for r in ranks}) != world:
            pg_info["warning"] = "two ranks report the same device"
            log(f"WARNING: two ranks share a device: {ranks}")
    else:
        pg_info["ranks"] = [me]
    # Preflight (N > 1 ranks, or a one-rank group with SIMSEG_FORCE_COLLECTIVES=1): every collective of the step once with a value check, then
    # a few timed repetitions each (the isolated rates of roofline.comm).  A wrong value, or two ranks on one device under nccl, ends the run
    # HERE with a JSON error line and a non-zero exit code - before anything is timed.
    comm_probe = None
    from simseg_amd.parallel import force_collectives
    if dist.is_initialized() and (world > 1 or force_collectives()):
        from simseg_amd import commcheck
        if os.environ.get("SIMSEG_BENCH_SABOTAGE_PREFLIGHT") == "1" and rank == world - 1:      # tests: this rank's all-reduce returns wrong values
            _real_all_reduce = dist.all_reduce

            def _broken_all_reduce(t, *a, **kw):
                w = _real_all_reduce(t, *a, **kw)
                if t.is_floating_point():
                    t.add_(1.0)
                return w
            dist.all_reduce = _broken_all_reduce
        try:
            comm_probe = commcheck.preflight(dev, args.pairs_per_gpu, ranks_info=pg_info["ranks"])
            log(f"communication preflight ok: { {k: (v.get('busbw_GBps'), v['ok']) for k, v in comm_probe.items()} }")
        except commcheck.PreflightError as e:
            log(f"communication preflight FAILED: {e}")
            if rank == 0:
                os.write(_JSON_FD, (json.dumps({"metric": "image-text pairs/sec (train) + seg images/sec (eval), ViT-B", "value": None, "unit": "pairs/s",
                                                "n_gpus": world, "error": f"communication preflight failed: {e}"[:800],
                                                "config": {"process_group": pg_info}}) + "\n").encode())
            try:
                dist.destroy_process_group()
            except Exception:       # noqa: BLE001
                pass
            sys.exit(3)
    from simseg.models import PIPELINE
    torch.manual_seed(1234)
    model = build(cfg.model.name, cfg, PIPELINE).to(dev).train()
    net = model
    sync = None
    # Gradient exchange for N > 1 (SIMSEG_BENCH_DP): "bucket" (DEFAULT since round 4) = simseg_amd.parallel.GradSync - bucketed RCCL
    # all-reduces enqueued behind the backward of BOTH tower streams, i.e. the same two-stream tower schedule as the N = 1 line, so that the
    # 1 -> N ratio measures communication and not a schedule change (tested equal to torch DDP's averaged gradients on 2 ranks);
    # "flat" = the same with one all-reduce after the backward; "ddp" = torch DDP, whose bucket hooks synchronise with ONE stream: the
    # towers then run on one stream (+5-6 ms per step at N = 1).
    dp = os.environ.get("SIMSEG_BENCH_DP", "bucket")
    # SIMSEG_BENCH_FORCE_SYNC=1: run the N > 1 exchange machinery (flat gradient buffer, per-parameter hooks, events, communication stream;
    # collectives are skipped at world size 1) on ONE rank - what the N > 1 schedule costs before any byte travels
    if world > 1 or os.environ.get("SIMSEG_BENCH_FORCE_SYNC", "0") == "1":
        if dp in ("flat", "bucket"):
            from simseg_amd.parallel import GradSync
            sync = GradSync(model.parameters(), overlap=(dp == "bucket"), average="defer")      # (the mean over ranks is applied by the AdamW kernel)
            os.environ.setdefault("SIMSEG_AMD_TWO_STREAMS", "1")
        else:
            net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local], gradient_as_bucket_view=True, bucket_cap_mb=128)
    HALF = {"fp16": torch.float16, "bf16": torch.bfloat16}
    opt = AdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=1e-3, half_dtype=HALF[HEAD])
    from simseg_amd.optim import GradScaler
    # torch.amp.GradScaler (its defaults, as the reference constructs it: scale 65536, back-off 0.5, growth every 2000 clean steps) with this
    # package's one-kernel overflow check; the skip decision stays on the device (no host read in the step).  With N > 1 ranks the check reads
    # the EXCHANGED gradients (an inf on any rank is an inf in every rank's sum), so every rank takes the same decision.
    scaler = GradScaler("cuda")
    mode = [HEAD]

    def set_mode(m):
        """Arithmetic of the step: "fp16" (the reference's AMP: fp16 compute, scaled loss, live GradScaler) or "bf16" (no scaler)."""
        if m != mode[0]:
            mode[0] = m
            os.environ["SIMSEG_AMD_COMPUTE"] = m
            opt.half_dtype = HALF[m]          # the optimizer kernel's 16-bit weight copies follow the compute type
            opt._plans.clear()
    # single process, two tower streams: the text tower's parameters are updated on the text tower's stream, i.e. as soon as ITS backward
    # has finished, beside the rest of the image tower's backward (SIMSEG_BENCH_OPT_STREAMS=0: one launch behind everything, as before)
    OPT_STREAMS = world == 1 and sync is None and os.environ.get("SIMSEG_BENCH_OPT_STREAMS", "1") != "0"
    if OPT_STREAMS:
        from simseg.models.pipelines.clip import _side_stream
        side_ = _side_stream(dev)
        opt.set_param_streams({p: side_ for p in list(model.text_encoder.parameters()) + list(model.text_projection.parameters())})
    B, L = args.pairs_per_gpu, args.seq_len
    # NB pre-generated device batches, visited round-robin; the caption tensors handed to the model are FRESH tensor objects every
    # step (a clone, as a loader delivers them), so whatever the model derives from a mask - towers.ragged_maps with its host read of
    # the real-token count, clip_k_to_shortest - is rebuilt inside the timed region every step instead of being cached on the tensor
    NB = max(1, args.batches)
    HOST_LENGTHS = os.environ.get("SIMSEG_BENCH_HOST_LENGTHS", "1") != "0"      # (0: the model reads the real-token count back from the device)
    batches = [synthetic_batch(B, args.img, L, 30522, 1000 + rank + 100 * i, dev) for i in range(NB)]
    step_no = [0]
    log(f"model and {NB} batches on device")

    def next_batch():
        b = batches[step_no[0] % NB]
        step_no[0] += 1
        out = {"image": b["image"], "input_ids": b["input_ids"].clone(), "attention_mask": b["attention_mask"].clone()}
        if HOST_LENGTHS:
            out["caption_lengths"] = b["caption_lengths"]
        return out

    ZERO_COPY = os.environ.get("SIMSEG_BENCH_SYNC_ZERO_COPY", "0") != "0"      # (measured 0.7 ms SLOWER than the bucket copies: profiles/r4_gradsync_events_ab.txt)

    def step():
        opt.zero_grad(set_to_none=(world == 1 or sync is not None))     # under DDP the grads are views into the all-reduce buckets
        if sync is not None and ZERO_COPY:
            sync.begin()                                                 # the large weight gradients are written straight into the exchange buffer
        loss_dict, _, _ = net(next_batch())
        amp = mode[0] == "fp16"
        (scaler.scale(loss_dict["nce_loss"]) if amp else loss_dict["nce_loss"]).backward()
        if sync is not None:
            sync()
        kw = dict(grad_scale=sync.grad_scale if sync is not None else 1.0, param_streams=bool(getattr(model, "two_streams_used", False)))
        if amp:
            scaler.step(opt, **kw)       # overflow check (one read-only kernel) + AdamW with the unscale and the skip decision inside the kernel
            scaler.update()
        else:
            opt.step(**kw)
        return loss_dict["nce_loss"]

    bdf = None
    try:
        pr = torch.cuda.get_device_properties(dev)
        bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
    except Exception:       # noqa: BLE001  (older torch: no PCI fields - rocm-smi then)
        pass
    clocks = ClockSampler(bdf=bdf) if rank == 0 else None      # clock / power from a side thread: the chip is power-capped under this load
    for i in range(args.warmup):
        step()
        if i == 0:
            torch.cuda.synchronize()
            log("first step done")
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    if clocks is not None:
        clocks.mark()
    taken0 = opt.steps_taken() if HEAD == "fp16" else 0        # (a host read, outside the timed region)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    clock_info = clocks.stop() if clocks is not None else None
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed)
    log(f"timed region: {elapsed:.3f}s for {args.steps} steps")
    amp_info = None
    if HEAD == "fp16":
        amp_info = {"loss_scale": scaler.get_scale(), "optimizer_steps_taken": opt.steps_taken() - taken0, "steps": args.steps,
                    "note": "GradScaler live: scaled backward; overflow check = one read-only kernel, unscale + skip decision inside the AdamW "
                            "kernel on the device (no host read in the step)"}

    # ---- the same step with the padded caption tokens computed, as HF's BertModel does (secondary figure, same process) ----------
    dense_text = None
    if FULL and os.environ.get("SIMSEG_AMD_PACKED_TEXT", "1") != "0":
        os.environ["SIMSEG_AMD_PACKED_TEXT"] = "0"
        try:
            step()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            el2 = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
            if world > 1:
                dist.all_reduce(el2, op=dist.ReduceOp.MAX)
            dense_text = {"pairs_per_s": round(world * B * args.steps / float(el2), 2), "ms_per_step": round(1e3 * float(el2) / args.steps, 3)}
        finally:
            os.environ["SIMSEG_AMD_PACKED_TEXT"] = "1"

    # ---- the same step with round 3's forms of two backward streams: the saved GELU' as a 16-bit image (default: the 8-bit image of
    # simseg_gemm act 7 / 8) and the ViT residual-stream gradient as an fp32 image between the LayerNorm backward kernels (default: their
    # 16-bit copies only).  Same gradient fidelity against the exact-fp32 backward (tests/test_gpu_fullsize.py: mean 1 - cosine 5.06e-4 vs
    # 5.04e-4).  Secondary figure, same process.
    gelu16 = None
    from simseg_amd import towers as _tw
    compact0 = (_tw._GELU8, _tw._RES16, _tw._XHAT_Y)
    if FULL and any(compact0) and os.environ.get("SIMSEG_BENCH_GELU16_LEG", "1") != "0":
        _tw._GELU8 = _tw._RES16 = _tw._XHAT_Y = False
        try:
            for _ in range(2):
                step()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            el4 = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
            if world > 1:
                dist.all_reduce(el4, op=dist.ReduceOp.MAX)
            gelu16 = {"pairs_per_s": round(world * B * args.steps / float(el4), 2), "ms_per_step": round(1e3 * float(el4) / args.steps, 3)}
        finally:
            _tw._GELU8, _tw._RES16, _tw._XHAT_Y = compact0

    # ---- the same step in the OTHER 16-bit type (secondary figure, same process): bf16 without a scaler when the headline is the reference's
    # fp16 AMP, and the other way round
    other_leg = None
    OTHER = "bf16" if HEAD == "fp16" else "fp16"
    if os.environ.get("SIMSEG_BENCH_OTHER_DTYPE_LEG", os.environ.get("SIMSEG_BENCH_FP16", "1")) != "0":
        set_mode(OTHER)
        try:
            for _ in range(max(3, args.warmup)):
                step()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            taken1 = opt.steps_taken() if OTHER == "fp16" else 0
            t2 = time.perf_counter()
            for _ in range(args.steps):
                step()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            el3 = torch.tensor([time.perf_counter() - t2], device=dev, dtype=torch.float64)
            if world > 1:
                dist.all_reduce(el3, op=dist.ReduceOp.MAX)
            other_leg = {"dtype": OTHER, "pairs_per_s": round(world * B * args.steps / float(el3), 2), "ms_per_step": round(1e3 * float(el3) / args.steps, 3)}
            if OTHER == "fp16":
                other_leg.update({"loss_scale": scaler.get_scale(), "optimizer_steps_taken": opt.steps_taken() - taken1, "steps": args.steps})
        finally:
            set_mode(HEAD)
            step()                           # (weight copies of the headline type are back before the instrumented step)

    # ---- the same step with every collective replaced by a local stand-in of the same shapes (N > 1 ranks, or forced collectives on one):
    # buffers, hooks, events and the communication stream stay, no byte travels.  step(N) - this = the communication the step did not hide.
    # Last of the training legs: without the exchange the ranks' weights drift apart, which nothing after it depends on.
    no_comm = None
    comm_sizes = (sync.flat.numel() * 4, len(sync.buckets)) if sync is not None else None
    if sync is not None and comm_probe is not None:
        from simseg_amd import heads as _heads
        _heads.LOCAL_STANDIN, sync.skip_collectives = True, True
        try:
            for _ in range(2):
                step()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            for _ in range(args.steps):
                step()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            el5 = torch.tensor([time.perf_counter() - t3], device=dev, dtype=torch.float64)
            if world > 1:
                dist.all_reduce(el5, op=dist.ReduceOp.MAX)
            no_comm = {"ms_per_step": round(1e3 * float(el5) / args.steps, 3), "pairs_per_s": round(world * B * args.steps / float(el5), 2)}
        finally:
            _heads.LOCAL_STANDIN, sync.skip_collectives = False, False
        step()

    # ---- roofline of the dominant kernel: one extra instrumented step, events around every GEMM launch -------------
    # The timed steps run the two towers on two HIP streams (their kernels share the GPU, so a per-kernel duration is not
    # that kernel's own speed); the instrumented step runs them on ONE stream so that each launch is timed alone.
    env_ts = os.environ.get("SIMSEG_AMD_TWO_STREAMS")
    two_streams = (env_ts != "0") if env_ts is not None else world == 1      # the default of simseg/models/pipelines/clip.py
    if sync is not None:
        two_streams = os.environ.get("SIMSEG_AMD_TWO_STREAMS") != "0"
    os.environ["SIMSEG_AMD_TWO_STREAMS"] = "0"
    ops.PROFILE = []
    prof_lens = batches[step_no[0] % NB]["attention_mask"].sum(1).cpu()      # caption lengths of the instrumented step's batch
    step()
    torch.cuda.synchronize()
    if env_ts is None:
        del os.environ["SIMSEG_AMD_TWO_STREAMS"]
    else:
        os.environ["SIMSEG_AMD_TWO_STREAMS"] = env_ts
    agg = {}
    for kind, fl, e0, e1 in ops.PROFILE:
        a = agg.setdefault(kind, [0, 0.0, 0.0])
        a[0] += 1; a[1] += fl; a[2] += e0.elapsed_time(e1) * 1e-3
    ops.PROFILE = None
    attn_roofline = attention_roofline(dev, B, args.img, L, dtype=HALF[HEAD]) if rank == 0 else {}
    if rank == 0:
        attn_roofline.update(guarded("layernorm roofline", layernorm_roofline, dev, B, args.img, dim, dtype=HALF[HEAD], all_ranks=False) or {})
    os.environ["SIMSEG_AMD_COMPUTE"] = "bf16"          # (the evaluation legs below choose their own type)
    del net, model, opt, batches
    torch.cuda.empty_cache()
    seg = None
    # N > 1 (the driver's scaling runs): the headline, and of the evaluation legs only BASELINE configs[3]'s data-parallel one (source images
    # sharded over the ranks); the rest is single-GPU material the N = 1 line carries (SIMSEG_BENCH_FULL=1: everything at every N)
    if not args.no_seg:
        seg = {}
        if FULL:
            seg = {"fp32": guarded("seg fp32", seg_eval_bench, dev, world, "fp32"), "bf16": guarded("seg bf16", seg_eval_bench, dev, world, "bf16", windows=256),
                   # the same stage with the reference's DenseCRF in (device mean field on permutohedral lattices, tools/seg_evaluation.py:153)
                   "fp32_crf": guarded("seg fp32 crf", seg_eval_bench, dev, world, "fp32", crf=True, steps=1),
                   "bf16_crf": guarded("seg bf16 crf", seg_eval_bench, dev, world, "bf16", crf=True, steps=1, windows=256),
                   "vit_s_288_fp32_crf": guarded("seg vit-s crf", seg_eval_bench, dev, world, "fp32", windows=64, img=288, classes=21, tag="vit_small_patch16_224_in21k", dim=384, crf=True, steps=1),
                   # BASELINE configs[1]: ViT-S, reference-faithful 288^2 input (324 patches), 21 VOC classes
                   "vit_s_288_fp32": guarded("seg vit-s fp32", seg_eval_bench, dev, world, "fp32", windows=64, img=288, classes=21, tag="vit_small_patch16_224_in21k", dim=384),
                   "vit_s_288_bf16": guarded("seg vit-s bf16", seg_eval_bench, dev, world, "bf16", windows=256, img=288, classes=21, tag="vit_small_patch16_224_in21k", dim=384),
                   # BASELINE configs[1] at its stated size: ViT-S on PASCAL-VOC-shaped 512 x 512 inputs (1024 patches, interpolated position embedding), 21 classes
                   "vit_s_512_fp32_crf": guarded("seg vit-s 512 crf", seg_eval_bench, dev, world, "fp32", windows=63, img=512, classes=21, tag="vit_small_patch16_224_in21k", dim=384, crf=True, steps=1),
                   "vit_s_512_fp32": guarded("seg vit-s 512 fp32", seg_eval_bench, dev, world, "fp32", windows=63, img=512, classes=21, tag="vit_small_patch16_224_in21k", dim=384),
                   "vit_s_512_bf16": guarded("seg vit-s 512 bf16", seg_eval_bench, dev, world, "bf16", windows=256, img=512, classes=21, tag="vit_small_patch16_224_in21k", dim=384)}
        # BASELINE configs[3] proper: 512 x 1024 source images through 3 overlapping windows each, stitched maps, images/s = SOURCE images
        seg["slide_512x1024"] = {"bf16": guarded("slide bf16", seg_slide_bench, dev, world, "bf16", images=256, steps=1, crf=False, window_batch=256),
                                 "bf16_crf": guarded("slide bf16 crf", seg_slide_bench, dev, world, "bf16", images=256, steps=1, crf=True, window_batch=256),
                                 "images_per_s_is": "source images per second over all ranks (SURVEY.md 8d cfg 4)"}
        if FULL:
            seg["slide_512x1024"]["fp32_crf"] = guarded("slide fp32 crf", seg_slide_bench, dev, world, "fp32", images=21, steps=1, crf=True)
        if rank == 0 and FULL:         # single-image latency (the reference tool's batch size), eager launches vs one hipGraph replay
            seg["latency_batch1"] = [seg_latency_bench(dev, "fp32", 288, 21, "vit_small_patch16_224_in21k", 384),
                                     seg_latency_bench(dev, "fp32", 512, 171, "vit_base_patch16_224_in21k", 768),
                                     seg_latency_bench(dev, "bf16", 512, 171, "vit_base_patch16_224_in21k", 768)]
        os.environ["SIMSEG_AMD_COMPUTE"] = "bf16"
        if cpu_seg is not None:
            seg["cpu_baseline"] = cpu_seg
        log(f"seg eval stage: {seg}")
    want_retr = (not args.no_seg) or args.retrieval
    retr_multi = guarded("retrieval multi-rank", retrieval_multi_rank_bench, dev, rank, world) if (world > 1 and want_retr and FULL) else None       # collective: every rank
    retr = retrieval_bench(dev) if (rank == 0 and want_retr and FULL) else None
    if retr is None and retr_multi is not None and rank == 0:
        retr = {"multi_rank": retr_multi}
    elif retr is not None and retr_multi is not None:
        retr["multi_rank"] = retr_multi
    if retr is not None and FULL:
        if not args.no_seg:
            retr["encoder_inclusive"] = retrieval_encode_bench(dev)
        if cpu_retr is not None:
            retr["cpu_baseline"] = cpu_retr
        os.environ["SIMSEG_AMD_COMPUTE"] = "bf16"
        log(f"retrieval eval: {retr}")

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.img, args.seq_len)
        if not args.no_seg:
            cpu_seg = cpu_baseline_seg()
            log(f"cpu seg baseline: {cpu_seg}")
            cpu_retr = cpu_baseline_retrieval()
            log(f"cpu retrieval baseline: {cpu_retr}")
            if seg is not None:
                seg["cpu_baseline"] = cpu_seg
            if retr is not None:
                retr["cpu_baseline"] = cpu_retr

    if rank == 0:
        n_patches = (args.img // 16) ** 2
        fpp = flops_per_pair(n_patches, dim, L)
        pairs = world * B * args.steps
        value = pairs / elapsed
        dom = max(agg, key=lambda k: agg[k][2])
        packed = os.environ.get("SIMSEG_AMD_PACKED_TEXT", "1") != "0" and os.environ.get("SIMSEG_AMD_SKIP_PAD_ROWS", "1") != "0"
        t_img = n_patches + 1
        attn_img = B * 12 * (dim // 64) * 4.0 * t_img * t_img * 64
        attn_txt = 12 * 12 * 4.0 * 64 * (float((prof_lens.double() ** 2).sum()) if packed else B * float(L) * L)
        exec_fl = sum(a[1] for a in agg.values()) + 3.0 * (attn_img + attn_txt)
        cnt, fl, sec = agg[dom]
        achieved = fl / sec / 1e12
        gemm_sec = sum(a[2] for a in agg.values())
        # HBM bytes per launch of that kernel: from the rocprofv3 PMC passes of THIS command on THIS round's kernels (counters cannot be
        # read from inside the process); the entry is used only if it was measured on the kernel variant that dominates now
        traffic, traffic_src = None, None
        blob = git_blob_sha1(os.path.join(REPO, "simseg_amd", "csrc", "gemm.hip"))
        try:
            with open(os.path.join(REPO, "profiles", TRAFFIC_JSON)) as f:
                tj = json.load(f)
            ent = tj["per_kind"].get(dom)
            if ent and tj.get("gemm_hip_blob") == blob:
                traffic = ent["hbm_bytes_per_launch"]
                traffic_src = (f"profiles/{TRAFFIC_JSON} (PMC passes taken at commit {tj.get('commit')}, gemm.hip blob {blob[:12]} = this tree's): "
                               f"FETCH_SIZE x2 + WRITE_SIZE of {ent['kernel'][:60]}")
            elif ent:
                traffic_src = (f"null: profiles/{TRAFFIC_JSON} was measured on gemm.hip blob {str(tj.get('gemm_hip_blob'))[:12]}, this tree has {blob[:12]} "
                               "(re-run tools/profile_round.sh + tools/pmc_traffic_json.py)")
        except (OSError, ValueError, KeyError):
            pass
        kdesc = {"_P": "gemm_pp_kernel (256x256 tile, two wave groups in ping-pong, direct-to-LDS half-tile ring)",
                 "_Q": "gemm_pp2_kernel (persistent 256x256 ping-pong: one workgroup per CU walks its XCD's tiles, operand copies and the epilogue's stores run across tile boundaries)",
                 "_L": "gemm_large_kernel (256x256 tile, direct-to-LDS ring)"}.get(dom[-2:], "gemm_kernel (128x128 tile, 32x32x16 bf16 MFMA)")
        # every GEMM class of the instrumented step (kind = operand layout _ output type _ kernel: P per-tile ping-pong, Q persistent
        # ping-pong, S small-problem, none = 128x128), the aggregate, and the attention kernels timed alone on the same shapes
        per_class = {k: {"launches": v[0], "ms": round(1e3 * v[2], 3), "tflops": round(v[1] / v[2] / 1e12, 1),
                         "frac": round(v[1] / v[2] / (PEAK_F32 if k.startswith("f32") else PEAK_BF16), 4)} for k, v in sorted(agg.items())}
        gfl = sum(v[1] for k, v in agg.items() if not k.startswith("f32"))
        gsec = sum(v[2] for k, v in agg.items() if not k.startswith("f32"))
        per_class["all_bf16_gemms"] = {"launches": sum(v[0] for k, v in agg.items() if not k.startswith("f32")), "ms": round(1e3 * gsec, 3),
                                       "tflops": round(gfl / gsec / 1e12, 1), "frac": round(gfl / gsec / PEAK_BF16, 4)}
        per_class.update(attn_roofline)
        comm_block = None
        if comm_probe is not None and comm_sizes is not None:
            from simseg_amd import commcheck
            comm_block = commcheck.comm_roofline(world, B, comm_sizes[0], comm_sizes[1], comm_probe, 1e3 * elapsed / args.steps,
                                                 no_comm["ms_per_step"] if no_comm else None, process_group=pg_info)
            comm_block["preflight"] = comm_probe

        def _pick(d, *path):
            for k in path:
                d = d.get(k) if isinstance(d, dict) else None
            return d
        # the figures a reader of the parsed line (metric / value / config / roofline only) should still see
        secondary = {f"train_pairs_per_s_{OTHER}": other_leg["pairs_per_s"] if other_leg else None,
                     "seg_eval_bf16_windows_per_s": _pick(seg, "bf16", "windows_per_s"),
                     "seg_eval_bf16_crf_windows_per_s": _pick(seg, "bf16_crf", "windows_per_s"),
                     "seg_eval_vit_s_512_bf16_windows_per_s": _pick(seg, "vit_s_512_bf16", "windows_per_s"),
                     "seg_slide_512x1024_bf16_images_per_s": _pick(seg, "slide_512x1024", "bf16", "images_per_s"),
                     "seg_slide_512x1024_bf16_crf_images_per_s": _pick(seg, "slide_512x1024", "bf16_crf", "images_per_s")}
        out = {
            "metric": "image-text pairs/sec (train) + seg images/sec (eval), ViT-B", "value": round(value, 2), "unit": "pairs/s",
            "value_is": "training image-text pairs/s over all ranks; the zero-shot-seg eval rate is reported in seg_eval", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": HEAD, "data": "synthetic",
            "dtype_is": ("fp16 compute with fp32 accumulation and fp32 master weights under a live GradScaler - the reference's own AMP arithmetic "
                         "(clip_runner.py:226-230, core/hooks/optimizer.py:73-82)" if HEAD == "fp16" else "bf16 compute, fp32 accumulation, fp32 master weights, no loss scaling"),
            "amp": amp_info,
            f"value_{OTHER}": other_leg["pairs_per_s"] if other_leg else None, f"ms_per_step_{OTHER}": other_leg["ms_per_step"] if other_leg else None,
            "config": {"workload": f"ViT-B/16 + BERT-base contrastive pretrain step (fwd + global InfoNCE + bwd + AdamW), "
                                   f"{B} pairs/GPU, {args.img}x{args.img} images, {L}-token captions (BASELINE configs[2], weak-scaled)",
                       "image_encoder": args.tag, "text_encoder": "bert-base-uncased", "global_batch": world * B,
                       "pairs_per_gpu": B, "seq_len": L, "img_size": args.img, "parallelism": f"dp{world}",
                       "process_group": pg_info, "batches_rotated": NB, "secondary_figures": secondary,
                       "caption_lengths": ("host-side token counts travel with the batch (no host read in the step)" if HOST_LENGTHS
                                           else "derived from the device mask (one host read per step)"),
                       "gradient_sync": ((f"simseg_amd.parallel.GradSync ({dp})" + ((" [forced on one rank: every collective issued on the one-rank " + str(pg_info["backend"]) + " group]" if os.environ.get("SIMSEG_FORCE_COLLECTIVES", "0") == "1"
                                                                                        else " [forced on one rank: no collective]") if world == 1 else "")) if sync is not None
                                         else ("none" if world == 1 else "torch DDP")),
                       "gradient_sync_detail": ({"zero_copy_weight_gradients": ZERO_COPY, "gradients_copied_per_step": sync.copied_last,
                                                 "events_per_step": sync.events_last, "buckets": len(sync.buckets)} if sync is not None else None),
                       "optimizer_streams": ("text tower's parameters updated on its own stream (starts when that tower's backward ends)" if OPT_STREAMS else "one launch on the main stream"),
                       "tower_streams": 2 if two_streams else 1,
                       "persistent_gemm_reserved_cus": int(os.environ.get("SIMSEG_GEMM_PP2_RESERVE", "0")),
                       "bert_dropout": 0.1, "optimizer": "AdamW (fused HIP kernel)",
                       "captions": "ragged, lengths U{8..L}" + ("; the padded token rows of the text tower are not computed (same loss and "
                                   "gradients as computing them: SIMSEG_AMD_PACKED_TEXT=0)" if os.environ.get("SIMSEG_AMD_PACKED_TEXT", "1") != "0" else "; padded token rows computed")},
            "roofline": {"bound": "mfma", "kernel": f"{kdesc} <{dom}>",
                         "achieved": round(achieved, 2), "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s",
                         "frac": round(achieved * 1e12 / PEAK_BF16, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "kernel_is": "the GEMM class with the largest summed time; every class, the aggregate and the attention kernels are in per_class",
                         "per_class": per_class,
                         "launches_per_step": cnt, "avg_launch_ms": round(1e3 * sec / cnt, 4),
                         "flops_per_launch_avg": fl / cnt,
                         "measured": "one instrumented step with both towers on one stream (each launch alone on the GPU); the timed "
                                     "steps overlap the two towers on two streams" if two_streams else "one instrumented step",
                         "clocks_during_timed_steps": clock_info, "measured_on": me, "comm": comm_block,
                         "frac_of_peak_at_measured_clock": (round(achieved / clock_info["dense_bf16_peak_at_this_clock_tflops"], 4)
                                                            if clock_info else None),
                         # what bounds these launches (measured once per round, not in this process): the same kernels on ZERO operands - same
                         # bytes moved, same instructions - reach 1433-1447 TFLOP/s at ~1.13 kW; on Gaussian operands 1121-1186 at the board's
                         # ~1.36 kW: the long-K launches sit at the power envelope, the kernel's own ceiling is 0.58 of the nominal peak
                         "limits": {"power_capped_on_real_operands": True, "same_launches_on_zero_operands_tflops": {"tn": 1433, "nn": 1447, "nt_k768": 1097},
                                    "same_launches_on_gaussian_operands_tflops": {"tn": 1121, "nn": 1186, "nt_k768": 1007},
                                    "source": "profiles/r4_gemm_power_probe.txt (tools/gemm_power_probe.py)"}},
            "step_model": {"algorithmic_tflop_per_rank_step": round(B * fpp / 1e12, 2),
                           "algorithmic_tflops_per_gpu": round(B * fpp / (elapsed / args.steps) / 1e12, 2),
                           "algorithmic_frac_of_bf16_peak": round(B * fpp / (elapsed / args.steps) / PEAK_BF16, 4),
                           # EXECUTED work: every GEMM launch of the instrumented step (2 M N K as launched, incl. the zero rows that pad
                           # the packed caption rows to full tiles) + the attention kernels' 4 T^2 64 per head forward and 2x that
                           # backward, captions counted up to their own length when the padded rows are skipped
                           "executed_tflop_per_rank_step": round(exec_fl / 1e12, 2),
                           "whole_step_tflops_per_gpu": round(exec_fl / (elapsed / args.steps) / 1e12, 2),
                           "whole_step_frac_of_bf16_peak": round(exec_fl / (elapsed / args.steps) / PEAK_BF16, 4),
                           "whole_step_frac_counts": "executed FLOPs (the algorithmic_* keys count the reference's padded caption tokens too)",
                           "gemm_time_share_single_stream": round(gemm_sec / (elapsed / args.steps), 3),
                           "gemm_breakdown_ms": {k: round(1e3 * v[2], 3) for k, v in sorted(agg.items())},
                           "final_loss": round(float(loss.detach()), 4),
                           "with_padded_caption_tokens_computed": dense_text, "with_round3_forms_of_the_backward_streams": gelu16,
                           "backward_streams": {"gelu_derivative_image": ("8-bit (simseg_gemm act 7 / 8: uniform grid over GELU''s range; the product it feeds is as "
                                                                          "accurate as with the 16-bit image)" if _tw._GELU8 else "16-bit"),
                                                "vit_residual_gradient_between_layernorm_backwards": "16-bit copies only" if _tw._RES16 else "fp32 image",
                                                "vit_layernorm_backward_normalised_value": "from the saved 16-bit output where the gains allow it" if _tw._XHAT_Y else "from the fp32 input",
                                                "fidelity": "mean 1 - cosine of every parameter gradient to the exact-fp32 backward, ViT-B + BERT-base, B = 256: 4.64e-4 "
                                                            "(all compact forms) vs 4.63e-4 (round-3 forms): tests/test_gpu_fullsize.py"},
                           "same_step_in_the_other_16_bit_type": other_leg},
            "seg_eval": seg,
            "retrieval_eval": retr,
            "cpu_baseline": cpu,
        }
        sys.stdout.flush()
        os.write(_JSON_FD, (json.dumps(out) + "\n").encode())
    if dist.is_initialized():
        if world > 1:
            dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
