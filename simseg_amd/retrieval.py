"""Multi-rank retrieval evaluation: the exchange step of tools/retrieval_evaluation.py:65-99 and the rank-0 metric of :25-63.

Every rank encodes its shard of the (image, caption) rows (DistributedSampler, simseg/datasets/clip/clip_dataset.py:222-225); the
per-rank embeddings and ids are all-gathered (`all_gather` of simseg/utils/dist.py:43-62 needs EQUAL shapes on every rank, so short
shards are padded with rows whose image_id is -1), the padding is dropped (`image_id > -1`, retrieval_evaluation.py:94-95) and rank 0
computes R@1/5/10 in both directions + RSUM from ONE similarity matrix (heads.retrieval_recalls_both through RetrievalMetric).
No collective besides that gather: the path shards by row."""
import torch
import torch.distributed as dist

KEYS = ("image_embeddings", "text_embeddings", "image_id", "caption_id")


def _on():
    return dist.is_available() and dist.is_initialized()


def pad_to_equal_count(shard, group=None):
    """Pad a rank's {image_embeddings [n,P], text_embeddings [n,P], image_id [n], caption_id [n]} with rows of zeros / id -1 up to
    the largest row count of any rank (one MAX all-reduce of a scalar): what the reference's equal-count loader guarantees."""
    n = shard["image_id"].shape[0]
    if not _on() or dist.get_world_size(group) == 1:
        return shard
    cnt = torch.tensor([n], device=shard["image_id"].device, dtype=torch.int64)
    dist.all_reduce(cnt, op=dist.ReduceOp.MAX, group=group)
    pad = int(cnt.item()) - n
    if pad == 0:
        return shard
    out = {}
    for k in KEYS:
        t = shard[k]
        fill = -1 if k in ("image_id", "caption_id") else 0
        out[k] = torch.cat([t, torch.full((pad,) + tuple(t.shape[1:]), fill, device=t.device, dtype=t.dtype)])
    return out


def gather_retrieval_sets(shard, group=None):
    """tools/retrieval_evaluation.py:89-95: all-gather the four per-rank tensors in rank order and drop the image_id = -1 rows."""
    from simseg.utils.dist import all_gather
    shard = pad_to_equal_count(shard, group)
    full = {k: torch.cat(all_gather(shard[k].contiguous(), group), 0) for k in KEYS}
    valid = full["image_id"] > -1
    return {k: v[valid] for k, v in full.items()}


def retrieval_metrics(collection, name="coco"):
    """The rank-0 body of calcaulate_retrieval_metrics_and_log (:25-63) for the image-text datasets: unique images vs all captions,
    both directions, percentages + RSUM.  Returns the summary dict."""
    from simseg.tasks.clip.hooks.utils import IndexedEmbInfo, RetrievalMetric
    retrieval = RetrievalMetric()
    img = IndexedEmbInfo("image", collection["image_id"], collection["image_embeddings"]).unique()
    txt = IndexedEmbInfo("text", collection["image_id"], collection["text_embeddings"])
    res = retrieval(img, txt)
    res.update(retrieval(txt, img))            # answered from the first call's similarity matrix
    summary = {}
    for k, v in res.items():
        k = k.replace("[image] to [text]", "I2T").replace("[text] to [image]", "T2I").replace(": ", "-")
        summary[k] = v * 100.0
    summary["RSUM"] = sum(summary.values())
    return {f"{name}_{k}": v for k, v in summary.items()}


def evaluate_sharded(shard, name="coco", group=None):
    """gather + metric: the summary dict on rank 0, None elsewhere (`@ENV.root_only`, :24)."""
    full = gather_retrieval_sets(shard, group)
    rank = dist.get_rank(group) if _on() else 0
    return retrieval_metrics(full, name) if rank == 0 else None
