"""Retrieval metric with the reference's call surface (simseg/tasks/clip/hooks/utils.py:9-75), computed by the fused
similarity GEMM + first-match-rank kernels instead of an M x N argsort and an int64 gid gather."""
from dataclasses import dataclass
from typing import Any, Dict

import torch

from simseg_amd.heads import retrieval_recalls_both

__all__ = ["IndexedEmbInfo", "EmbANN", "RetrievalMetric"]


@dataclass
class IndexedEmbInfo:
    emb_name: str
    group_idx: torch.Tensor  # [N]
    emb_mat: torch.Tensor    # [N, D]

    def unique(self):
        """One row per group id, keeping the last row of every run after a sort by id (utils.py:14-19)."""
        gid, order = torch.sort(self.group_idx)
        uni, count = torch.unique_consecutive(gid, return_counts=True)
        last = torch.cumsum(count, 0) - 1
        return IndexedEmbInfo(self.emb_name, uni, self.emb_mat[order][last])

    def to_chunks(self, chunk_size):
        for s in range(0, self.emb_mat.shape[0], chunk_size):
            yield IndexedEmbInfo(self.emb_name, self.group_idx[s:s + chunk_size], self.emb_mat[s:s + chunk_size])


class EmbANN:
    """The reference's exhaustive nearest-neighbour helper (utils.py:30-50) with its outputs: for every left row the right group ids sorted
    by descending similarity [M, N] and the matrix of matches with the row's own id.  RetrievalMetric above never materialises these (it
    counts strictly larger scores instead: two passes over the similarity matrix, no argsort, no int64 gather); this class is for callers
    that want the matrices themselves.  The similarity product runs on the fp32 MFMA GEMM when the embeddings live on the GPU; equal scores
    are ordered by column (a stable sort) where torch.argsort leaves the order unspecified.  With a chunk size the left rows go through in
    chunks and the two outputs are concatenated (the reference's `torch.cat` of a list of TUPLES, utils.py:47-50, cannot run)."""

    def __init__(self, chunk_size=None) -> None:
        self.chunk_size = chunk_size

    def _ann(self, leftemb: IndexedEmbInfo, rightemb: IndexedEmbInfo):
        a, b = leftemb.emb_mat, rightemb.emb_mat
        if a.is_cuda:
            from simseg_amd import ops
            sim = ops.gemm(a.float().contiguous(), b.float().contiguous())          # [M, N] = a . b^T
        else:
            sim = a.float() @ b.float().T
        order = torch.argsort(sim, dim=1, descending=True, stable=True)
        right_sorted = rightemb.group_idx.to(order.device)[order]                   # = gather(expand(right_gid), 1, order)
        matched = right_sorted == leftemb.group_idx.to(order.device).unsqueeze(1)
        return right_sorted, matched

    def __call__(self, leftemb: IndexedEmbInfo, rightemb: IndexedEmbInfo):
        if self.chunk_size is None:
            return self._ann(leftemb, rightemb)
        parts = [self._ann(sub, rightemb) for sub in leftemb.to_chunks(self.chunk_size)]
        return torch.cat([p[0] for p in parts], dim=0), torch.cat([p[1] for p in parts], dim=0)


class RetrievalMetric:
    def __init__(self, with_prefix=True):
        self.recall_range = (1, 5, 10)
        self.with_prefix = with_prefix
        self._reverse = None       # (key of the swapped call, its tensors kept alive, its result)

    @staticmethod
    def _key(*tensors):
        return tuple((t.data_ptr(), t._version, tuple(t.shape), t.dtype, t.device) for t in tensors)

    def __call__(self, leftemb: IndexedEmbInfo, rightemb: IndexedEmbInfo) -> Dict[str, Any]:
        # The evaluation calls the metric twice with swapped arguments (tools/retrieval_evaluation.py:44-45).  Both directions come from ONE
        # similarity matrix (rows rank their columns, columns rank their rows - heads.retrieval_recalls_both), so the first call also
        # produces the swapped call's answer and keeps it, keyed on the identity and version of the four tensors; the second call then
        # launches nothing.  An entry is used once; tensors modified in place or replaced in between miss the key and are recomputed.
        tensors = (leftemb.emb_mat, leftemb.group_idx, rightemb.emb_mat, rightemb.group_idx)
        key = self._key(*tensors)
        memo, self._reverse = self._reverse, None
        if memo is not None and memo[0] == key and all(a is b for a, b in zip(memo[1], tensors)):
            res = memo[2]
        else:
            res, rev = retrieval_recalls_both(*tensors, self.recall_range)
            swapped = (rightemb.emb_mat, rightemb.group_idx, leftemb.emb_mat, leftemb.group_idx)
            self._reverse = (self._key(*swapped), swapped, rev)
        if self.with_prefix:
            prefix = f"[{leftemb.emb_name}] to [{rightemb.emb_name}]:"
            res = {f"{prefix} {k}": v for k, v in res.items()}
        return res
