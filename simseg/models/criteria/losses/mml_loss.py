"""InfoNCE of the SimSeg contrastive objective (mirror of simseg/models/criteria/losses/mml_loss.py:12-103) on the
fused HIP loss path: fp32 MFMA similarity block -> in-place cross-entropy rows -> gradient GEMMs, with the embedding
exchange as an RCCL all-gather forward / reduce-scatter backward."""
import torch
import torch.distributed as dist
import torch.nn as nn

from simseg.utils import ENV, GatherLayer, logger
from simseg.utils.dist import generate_local_groups
from simseg_amd.heads import ClipLossFn, NCEFn, all_gather_rows

from .builder import LOSS

__all__ = ["NCE"]


@LOSS.register_obj
class NCE(nn.Module):
    def __init__(self, cfg, rank):
        super().__init__()
        self.cfg = cfg
        self.global_reduce = cfg.loss.global_reduce
        self.rank, self.group = 0, None
        self.gather_backward = False
        if self.global_reduce:
            self.gather_backward = cfg.loss.nce_loss.gather_backward
            if dist.is_available() and dist.is_initialized():
                group_size = cfg.loss.group_size if cfg.loss.group_size >= 0 else ENV.size
                self.group, self.rank = generate_local_groups(group_size)
                logger.info("NCE Loss Group size, Group Rank, Env Rank:", group_size, self.rank, ENV.rank, root_only=False)
            else:   # single process: the gather is the identity (the reference insists on a process group)
                logger.info("NCE: no process group, global_reduce degenerates to the local batch")
        t = torch.ones([]) * cfg.loss.temperature.value
        if cfg.loss.temperature.name == "parameter":
            self.temperature = nn.Parameter(t)
        elif cfg.loss.temperature.name == "constant":
            self.register_buffer("temperature", t, persistent=False)
        else:
            raise NotImplementedError(cfg.loss.temperature.name)
        self.smoothing = float(cfg.loss.smoothing)

    def _gather(self, t):
        if self.group is None:
            return t
        if self.gather_backward and t.requires_grad:
            return GatherLayer.apply(t, self.group, self.rank)
        return all_gather_rows(t, self.group)

    def both(self, image_embeddings, text_embeddings):
        """0.5 * (self(image, text)[0] + self(text, image)[0]) and the two top-1 accuracies - what CLIPModel.forward_loss computes with
        two calls of this module (pipelines/clip.py:129-140) - as ONE autograd node with the embedding exchange inside
        (simseg_amd.heads.ClipLossFn: same kernels, a third of the launches).  global_reduce losses without an ignore mask only."""
        if not self.global_reduce:
            raise NotImplementedError("NCE.both: the fused head is the global_reduce form")
        return ClipLossFn.apply(image_embeddings, text_embeddings, self.temperature, self.group, self.rank, self.smoothing, self.gather_backward)

    def forward(self, feat1, feat2, label=None, ignore_mask=None):
        if self.global_reduce:
            feat2_global = self._gather(feat2)
            ignore_global = None if ignore_mask is None else all_gather_rows(ignore_mask.float(), self.group) if self.group else ignore_mask.float()
            if feat2_global.shape[0] % feat1.shape[0] != 0:
                raise AssertionError(f"global size: {feat2_global.shape[0]}, batch size: {feat1.shape[0]}")
            return NCEFn.apply(feat1, feat2_global, self.temperature, ignore_mask, ignore_global, self.rank, self.smoothing)
        if ignore_mask is not None:
            raise NotImplementedError("ignore_mask with loss.global_reduce=False is outside the accelerated path")
        # local branch (:79-86): symmetric loss over the local batch, three return values
        l1, a1 = NCEFn.apply(feat1, feat2, self.temperature, None, None, 0, self.smoothing)
        l2, a2 = NCEFn.apply(feat2, feat1, self.temperature, None, None, 0, self.smoothing)
        return 0.5 * (l1 + l2), a1, a2
