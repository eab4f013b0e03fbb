"""Per-image body of the reference's `evaluate_benchmark` (tools/seg_evaluation.py:99-170) as batched device work.

    feats  = model.forward_image_feature(image)              # [B, n*n, D]
    pooled = model.forward_image_project(feats)              # [B, 512]
    sim    = patch_text_similarity(model.image_projection(feats), text)      # [B, n*n, C]   (K14)
    out    = segment(sim, pooled @ text.T, labels, n, top_cls_num)

Between the normalised map and the morphology the reference runs a fully connected CRF on the CPU for every visited candidate
(pydensecrf, :31-54 / :153).  Here it is `ops.dense_crf` - mean-field inference on two permutohedral lattices built on the device -
when the caller hands over the de-normalised network inputs (`images_u8`, what the tool builds at :104); without them the binary
map is the CRF's unary decision (prob > 0.5), and a caller-supplied `refine` hook can replace either.  Everything after it (7x7
dilate + erode, nearest resize, score-weighted argmax, IoU histograms) is the reference's arithmetic again.
"""
import os

import torch

from . import ops


CRF_PARAMS = dict(sxy_g=3.0, compat_g=3.0, sxy_b=40.0, srgb=13.0, compat_b=10.0, iters=3)        # tools/seg_evaluation.py:48-51


def crf_masks(prob, cand_idx, images_u8, chunk=None, scale=1, static=None, **params):
    """prob [B,K,h,w] fp32 normalised candidate maps (at 1/scale of the image resolution: the x16 nearest upsampling of the tool,
    tools/seg_evaluation.py:141-143, is applied here per chunk and only to visited slots), cand_idx [B,K] (-1 = slot not visited),
    images_u8 [B,H,W,3] -> uint8 masks [B,K,H,W] (0/255; unvisited slots zero).  One host read of the candidate table per batch (the
    reference loops on the host per image and candidate); images with a visited candidate go through the device CRF `chunk` at a time
    (an image's candidates share its lattices; images are grouped by their number of visited candidates, so chunks carry few idle maps).
    Per chunk: one gather of the visited low-resolution maps, one upsampling, one CRF call, one scatter of the masks."""
    B, K, h, w = prob.shape
    H, W = h * scale, w * scale
    dev = prob.device
    if static is None:
        static = os.environ.get("SIMSEG_CRF_STATIC", "0") != "0"
    if static:
        # Sync-free form (opt-in): no host read and no data-dependent shapes - every image goes through the CRF in ONE call with all K
        # candidate slots as channels; slots the reference never visits carry the all-zero map seg_masks leaves there and their result is
        # dropped.  K = 5 channels instead of the ~2 visited maps per image: measured 55.6 vs 28.4 ms per 63-window batch of 512^2
        # (round 3), so the chunked form below stays the default.  (Replaying the CRF inside a hipGraph is NOT supported: a first
        # attempt hung in replay - the hash build's probe loops never terminate if a table memset is not replayed - and was not pursued.)
        up = prob if scale == 1 else prob.repeat_interleave(scale, 2).repeat_interleave(scale, 3)
        m, _ = ops.dense_crf(images_u8.contiguous(), up.contiguous(), **dict(CRF_PARAMS, **params))
        return m * (cand_idx >= 0).to(torch.uint8)[:, :, None, None]
    if chunk is None:
        # images per call: ~16.7 M pixels (64 windows of 512^2, 32 images of 512 x 1024; 17.7 GB of workspace).  Round 5, after the lattice
        # build became cheaper per image: 16 / 32 / 64 windows of 512^2 per call = 20.7 / 19.9 / 19.1 ms per 63-window batch
        chunk = int(os.environ.get("SIMSEG_CRF_CHUNK", "0"))
        if not chunk:
            # ... bounded by what the device has FREE: the workspace is ~1.06 KB per pixel and call (17.7 GB at 16.7 M pixels), the encoder
            # pipeline keeps two batches in flight beside it, and a 2048^2 source image alone is 4.2 M pixels - so no floor above one image
            budget = 1 << 24
            if dev.type == "cuda":
                try:
                    free, _ = torch.cuda.mem_get_info(dev)
                    budget = min(budget, int(0.5 * free / 1100))
                except Exception:       # noqa: BLE001
                    pass
            chunk = max(1, min(128, budget // (H * W)))
    masks = torch.zeros(B, K, H, W, device=dev, dtype=torch.uint8)
    visited = (cand_idx >= 0).cpu()
    kw = dict(CRF_PARAMS, **params)
    todo = [(b, visited[b].nonzero().flatten().tolist()) for b in range(B)]
    todo = sorted(((b, ks) for b, ks in todo if ks), key=lambda t: len(t[1]))      # chunks of equal width: no idle zero maps
    for s in range(0, len(todo), chunk):
        part = todo[s:s + chunk]
        cmax = max(len(ks) for _, ks in part)
        slot = torch.full((len(part), cmax), -1, dtype=torch.int64)
        for j, (_, ks) in enumerate(part):
            slot[j, :len(ks)] = torch.tensor(ks)
        valid = (slot >= 0).to(dev)
        bsel = torch.tensor([b for b, _ in part], device=dev)
        ksel = slot.clamp(min=0).to(dev)                           # idle slots repeat a visited map (their result is dropped)
        lo = prob[bsel[:, None], ksel]                             # [n, cmax, h, w]
        up = lo if scale == 1 else lo.repeat_interleave(scale, 2).repeat_interleave(scale, 3)
        m, _ = ops.dense_crf(images_u8[bsel].contiguous(), up.contiguous(), **kw)
        masks[bsel[:, None].expand(-1, cmax)[valid], ksel[valid]] = m[valid]
    return masks


def segment_begin(sim, scores, num_patch, top_cls_num, ncand=5, need_prob=False):
    """First half of segment(): candidate selection and the per-candidate min-max maps - device work only, nothing is read back."""
    cand_idx, cand_score, thr = ops.seg_select(scores, top_cls_num, ncand)
    masks, prob = ops.seg_masks(sim, cand_idx, num_patch, want_prob=need_prob)
    return {"cand_idx": cand_idx, "cand_score": cand_score, "threshold": thr, "masks": masks, "prob": prob, "num_patch": num_patch,
            "num_classes": sim.shape[2]}


def segment_finish(st, labels, num_classes=None, ignore_index=255, refine=None, hist=None, want_pred=True, closing=True, images_u8=None):
    """Second half of segment(): DenseCRF (ONE host read of the candidate table), closing, resize + argmax + IoU histograms."""
    cand_idx, cand_score, masks, prob, num_patch = st["cand_idx"], st["cand_score"], st["masks"], st["prob"], st["num_patch"]
    num_classes = num_classes or st["num_classes"]
    if refine is not None or images_u8 is not None:
        B, K, N = prob.shape
        nh, nw = (num_patch, num_patch) if isinstance(num_patch, int) else num_patch
        lo = prob.view(B, K, nh, nw)
        if refine is not None:
            up = lo.repeat_interleave(16, 2).repeat_interleave(16, 3)
            masks = refine(up, cand_idx, cand_score).to(torch.uint8).contiguous()
        else:
            masks = crf_masks(lo, cand_idx, images_u8, scale=16)
    if closing:
        masks = ops.close7(masks, cand_idx.reshape(-1))                       # cv2.dilate then cv2.erode (:156-157), visited slots only
    pred, hist = ops.seg_predict(masks, cand_idx, cand_score, labels, num_classes, ignore_index, hist=hist, want_pred=want_pred)
    return {"pred": pred, "hist": hist, "cand_idx": cand_idx, "cand_score": cand_score, "threshold": st["threshold"], "masks": masks}


def segment(sim, scores, labels, num_patch, top_cls_num, num_classes=None, ncand=5, ignore_index=255, refine=None, hist=None,
            want_pred=True, closing=True, images_u8=None):
    """sim [B,n*n,C] fp32, scores [B,C], labels [B,H,W] uint8 -> dict(pred, hist, cand_idx, cand_score, threshold).
    images_u8 [B,16n,16n,3] uint8 (RGB): run the reference's DenseCRF on every visited candidate map."""
    st = segment_begin(sim, scores, num_patch, top_cls_num, ncand, need_prob=refine is not None or images_u8 is not None)
    return segment_finish(st, labels, num_classes or sim.shape[2], ignore_index, refine, hist, want_pred, closing, images_u8)


def encode_batch(model, image, text, top_cls_num, crf=True, mean=None, std=None, sim_dtype=None, refine=None):
    """Everything of eval_batch() up to the candidate maps: towers -> pooled embedding + projected patch tokens -> similarity map for all
    classes -> candidate selection + min-max maps.  Device work only (no host read): the caller may enqueue the NEXT batch's encoder
    before it finishes this one (EvalPipeline)."""
    from .heads import patch_text_similarity
    feats = model.forward_image_feature(image)                    # [B, n*n, D]
    pooled = model.forward_image_project(feats)                   # [B, 512]
    n = int(round(feats.shape[1] ** 0.5))
    sim = patch_text_similarity(model.image_projection(feats), text, compute_dtype=sim_dtype)
    raw = None
    if crf and refine is None:
        raw = (((image * std) + mean) * 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
    st = segment_begin(sim, ops.gemm(pooled.float(), text), n, top_cls_num, need_prob=refine is not None or raw is not None)
    st["images_u8"] = raw
    return st


# ---- sliding-window evaluation (BASELINE configs[3]: "ViT-B zero-shot seg eval, COCO-Stuff-shaped 171 classes, slide-window 512x512") ----
# The reference tool resizes every image to ONE network input (tools/seg_evaluation.py:84-85,109); SURVEY.md 8d cfg 4 defines the
# sliding-window form this package adds: windows of `win` pixels at stride `stride` over the (larger) source image, each window through the
# towers as an image of its own, the per-window patch x class similarity maps overlap-AVERAGED on the source image's patch grid, and then the
# reference's per-image body (candidate classes, min-max, DenseCRF, 7x7 closing, resize, score-weighted argmax, IoU areas) ONCE per source
# image on the stitched map.  Image-level class scores (:112-123 works on the pooled embedding of the one resized image) are the mean of
# the windows' scores.  oracle/segpost_ref.py sliding_window_image() is the per-image loop this is tested against.

def window_grid(H, W, win=512, stride=256):
    """-> (wy, wx): windows per column / row.  The image must be tiled exactly: (H - win) and (W - win) multiples of the stride (the
    caller pads or resizes otherwise), win and stride multiples of the 16-pixel patch."""
    if win % 16 or stride % 16 or stride <= 0 or stride > win:
        raise ValueError(f"window_grid: win {win} and stride {stride} must be multiples of the 16-pixel patch with 0 < stride <= win")
    if H < win or W < win or (H - win) % stride or (W - win) % stride:
        raise ValueError(f"window_grid: a {H}x{W} image is not tiled exactly by {win}-pixel windows at stride {stride}")
    return (H - win) // stride + 1, (W - win) // stride + 1


def extract_windows(image, win=512, stride=256):
    """image [B,3,H,W] -> [B*wy*wx, 3, win, win]: an image's windows consecutive, row-major over its window grid (a strided copy)."""
    B, C, H, W = image.shape
    wy, wx = window_grid(H, W, win, stride)
    v = image.unfold(2, win, stride).unfold(3, win, stride)            # [B, C, wy, wx, win, win] view
    return v.permute(0, 2, 3, 1, 4, 5).reshape(B * wy * wx, C, win, win)


def encode_batch_sliding(model, image, text, top_cls_num, win=512, stride=256, crf=True, mean=None, std=None, sim_dtype=None, window_batch=None):
    """encode_batch() for source images larger than the network input: image [B,3,H,W] -> B*wy*wx windows through the towers (at most
    `window_batch` windows per call), similarity maps stitched on the [H/16, W/16] patch grid (ops.stitch_windows), image-level scores =
    the mean of the windows' pooled scores -> candidate selection and min-max maps on the stitched grid.  Device work only."""
    from .heads import patch_text_similarity
    B, _, H, W = image.shape
    wy, wx = window_grid(H, W, win, stride)
    wins = extract_windows(image, win, stride)
    n = win // 16
    sims, scores = [], []
    wb = window_batch or wins.shape[0]
    for s in range(0, wins.shape[0], wb):
        feats = model.forward_image_feature(wins[s:s + wb])          # [b, n*n, D]
        pooled = model.forward_image_project(feats)                   # [b, 512]
        sims.append(patch_text_similarity(model.image_projection(feats), text, compute_dtype=sim_dtype))
        scores.append(ops.gemm(pooled.float(), text))
    sim_w = sims[0] if len(sims) == 1 else torch.cat(sims)
    sc_w = scores[0] if len(scores) == 1 else torch.cat(scores)
    sim = ops.stitch_windows(sim_w.float(), wy, wx, n, stride // 16)                       # [B, nh*nw, C]
    sc = ops.stitch_windows(sc_w.float().view(B * wy * wx, 1, -1), wy, wx, 1, 0).view(B, -1)      # mean over an image's windows
    raw = None
    if crf:
        raw = (((image * std) + mean) * 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
    st = segment_begin(sim, sc, (H // 16, W // 16), top_cls_num, need_prob=raw is not None)
    st["images_u8"] = raw
    st["windows"] = (wy, wx)
    return st


def shard_batches(batches, rank, world):
    """Round-robin shard of an iterable of batches: rank r takes batches r, r + world, ... (independent units, no data-path collective)."""
    for i, b in enumerate(batches):
        if i % world == rank:
            yield b


def evaluate_sharded(model, batches, text, top_cls_num, num_classes=None, group=None, slide=None, crf=True, mean=None, std=None, sim_dtype=None,
                     device=None, pipelined=None, window_batch=None, presharded=False):
    """The zero-shot segmentation evaluation over `batches` = an iterable of (image [b,3,H,W], label [b,Hl,Wl] uint8) - the SAME iterable on
    every rank - data-parallel over the ranks of `group` (default: the world, or a single process when torch.distributed is not
    initialised): batches are dealt round-robin (shard_batches; the reference's loader gives every rank every image,
    simseg/datasets/seg/seg_dataset.py:67-81), each rank accumulates its [3, C] area histograms on its device and ONE all-reduce(SUM) of that
    tensor ends the evaluation (simseg/utils/metrics.py:85-97 sums the same three vectors over the images).  Every rank still ITERATES the
    whole iterable (a batch it skips is produced and dropped): when producing a batch is expensive, hand in this rank's share only - a loader
    over a dataset sharded with the same i % world == rank rule - and say presharded=True (tools/seg_eval_device.py does).  slide = (win, stride): the
    sliding-window form (encode_batch_sliding); None: one network input per image (encode_batch).
    -> dict(iou [C] float64, miou, hist [3,C] int64 (global), images (global count), images_local)."""
    import torch.distributed as dist
    on = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank(group) if on else 0
    world = dist.get_world_size(group) if on else 1
    C = num_classes or text.shape[0]
    dev = torch.device(device) if device is not None else text.device
    hist = torch.zeros(3, C, device=dev, dtype=torch.int64)

    def encode(image, label):
        if slide is not None:
            return encode_batch_sliding(model, image, text, top_cls_num, win=slide[0], stride=slide[1], crf=crf, mean=mean, std=std,
                                        sim_dtype=sim_dtype, window_batch=window_batch)
        return encode_batch(model, image, text, top_cls_num, crf=crf, mean=mean, std=std, sim_dtype=sim_dtype)

    def finish(st, image, label):
        return finish_batch(st, label, hist=hist)

    count = 0
    pipe = EvalPipeline(dev, encode, finish, pipelined=crf if pipelined is None else pipelined)
    with torch.no_grad():
        for image, label in (batches if presharded else shard_batches(batches, rank, world)):      # presharded: `batches` is already this rank's share
            pipe.submit(image.to(dev, non_blocking=True), label.to(dev, non_blocking=True))
            count += image.shape[0]
        pipe.flush()
    n_img = torch.tensor([count], device=dev, dtype=torch.int64)
    if world > 1:
        dist.all_reduce(hist, op=dist.ReduceOp.SUM, group=group)          # the evaluation's only collective: 3*C int64 areas
        dist.all_reduce(n_img, op=dist.ReduceOp.SUM, group=group)
    iou, miou = iou_from_hist(hist)
    return {"iou": iou, "miou": miou, "hist": hist, "images": int(n_img), "images_local": count, "rank": rank, "world": world}


def finish_batch(st, label, hist=None, refine=None, want_pred=False):
    return segment_finish(st, label, hist=hist, want_pred=want_pred, refine=refine, images_u8=st.get("images_u8"))


def eval_batch(model, image, label, text, top_cls_num, hist=None, crf=True, mean=None, std=None, sim_dtype=None, refine=None,
               want_pred=False):
    """One batch of the zero-shot segmentation evaluation, everything on the device (tools/seg_evaluation.py:99-170 for every image of
    the batch at once): towers -> pooled embedding + projected patch tokens -> similarity map for all classes -> segment().
    image [B,3,S,S] normalised network input, label [B,H,W] uint8, text [C,512] unit-norm class embeddings.  crf: run the DenseCRF on
    the de-normalised input (image * std + mean, :104) as the reference does; hist [3,C] int64 accumulates."""
    st = encode_batch(model, image, text, top_cls_num, crf=crf, mean=mean, std=std, sim_dtype=sim_dtype, refine=refine)
    return finish_batch(st, label, hist=hist, refine=refine, want_pred=want_pred)


_PIPE_STREAMS = {}


def _pipeline_streams(device):
    """Two normal-priority encoder streams + one high-priority finishing stream per device, created ONCE per process: the runtime maps
    streams onto a few hardware queues when they are created, and two streams that land on one queue serialise - a fresh set per
    evaluation made the finishing stage's luck (its own queue, or behind an encoder's) vary from call to call."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    st = _PIPE_STREAMS.get(key)
    if st is None:
        lo, hi = 0, -1
        try:
            lo, hi = torch.cuda.Stream.priority_range()
        except Exception:       # noqa: BLE001  (older torch: the documented default range)
            pass
        st = _PIPE_STREAMS[key] = ([torch.cuda.Stream(device=device, priority=lo) for _ in range(2)], torch.cuda.Stream(device=device, priority=hi))
    return st


class EvalPipeline:
    """Software-pipelined evaluation: batch i's ENCODER is enqueued before batch i-1 is finished, and the finishing stage runs on its
    own HIGH-PRIORITY stream.  The DenseCRF stage needs one host read (the candidate table) and then issues ~100 small dependent
    launches per chunk from a Python loop; run back to back with its own encoder, the GPU idles through that loop (6 of the 23 ms the
    stage took per 63-window batch of 512^2).  Here the encoders alternate between two normal-priority streams (two batches in flight:
    one's attention / LayerNorm phases fill the tile-grid tails of the other's GEMMs), so MFMA work is always queued, and the stage's
    small kernels take the CUs they need as soon as a GEMM tile retires (without the priority a chain of small dependent kernels
    queues behind whole GEMM launches of the other stream: measured 3x slower end to end in fp32).  Same work per batch, same results
    (the histograms accumulate atomically: no ordering between batches is needed).
    Memory: a batch's intermediate tensors are produced under an encoder stream and read by the finishing stream.  They are kept
    referenced here until an event behind the finishing stage has completed (no record_stream: that defers the allocator's reuse of
    every marked block and sent each batch back to hipMalloc), at most `depth` batches deep."""

    def __init__(self, device, encode, finish, depth=3, pipelined=True):
        # pipelined = False: whole batches on the two alternating streams, each finished right behind its encoder - the better order when
        # the finishing stage has no host read and no long chain of small launches (no DenseCRF: 3071 vs 2927 windows/s, ViT-B bf16 512^2)
        self.pipelined = pipelined
        self.enc_streams, self.post_stream = _pipeline_streams(torch.device(device))
        self.encode, self.finish, self.depth = encode, finish, depth
        self.i, self.pending, self.last, self.retire = 0, None, None, []

    def _reap(self, keep):
        while self.retire and (len(self.retire) > keep or self.retire[0][0].query()):
            self.retire[0][0].synchronize()
            self.retire.pop(0)

    def submit(self, *batch):
        self._reap(self.depth)
        cur = torch.cuda.current_stream()
        st = self.enc_streams[self.i % 2]
        self.i += 1
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            enc = self.encode(*batch)
            ev = torch.cuda.Event()
            ev.record()
            if not self.pipelined:
                self.last = self.finish(enc, *batch)
                # the batch was allocated under the caller's stream and is read by kernels queued on this one: keep it (and the encoder's
                # outputs) referenced until an event behind them has completed, as the pipelined path does - otherwise the caching
                # allocator may hand those blocks to the caller's next host->device copy while they are still being read
                done = torch.cuda.Event()
                done.record()
                self.retire.append((done, enc, batch))
                return
        prev, self.pending = self.pending, (ev, enc, batch)
        if prev is not None:
            self._finish(prev)

    def _finish(self, item):
        ev, enc, batch = item
        self.post_stream.wait_event(ev)
        with torch.cuda.stream(self.post_stream):
            self.last = self.finish(enc, *batch)
            done = torch.cuda.Event()
            done.record()
        self.retire.append((done, enc, batch))

    def flush(self):
        if self.pending is not None:
            self._finish(self.pending)
            self.pending = None
        self._reap(0)
        cur = torch.cuda.current_stream()
        for st in self.enc_streams + [self.post_stream]:
            cur.wait_stream(st)
        return self.last


def iou_from_hist(hist):
    """hist [3,C] int64 -> (per-class IoU float64 with NaN where the class never occurs, mean over non-NaN) as :172-173."""
    inter, pred, label = hist[0].double(), hist[1].double(), hist[2].double()
    union = pred + label - inter
    iou = inter / union
    return iou, iou[~torch.isnan(iou)].mean()
