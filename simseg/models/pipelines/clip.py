"""The drop-in boundary: `CLIPModel` with the reference's attribute / method / state-dict surface
(simseg/models/pipelines/clip.py:13-229), computing on the MI355X-native towers and heads.

state-dict keys: image_encoder.model.model.* (timm names), text_encoder.model.model.* (HF names),
image_projection.linear.weight, text_projection.linear.weight, loss.temperature."""
import numpy as np
import os

import torch
import torch.nn as nn

from simseg.models.backbones.builder import BACKBONE
from simseg.models.criteria.losses.builder import LOSS
from simseg.models.pipelines.builder import PIPELINE
from simseg.utils import ENV, logger
from simseg_amd.heads import discard_prefetched, prefetch_gather
from simseg_amd.nn import compute_dtype
from simseg_amd.towers import ProjectPoolFn, packed_text

from ..components import AvgPooling, ComplexProjection, L2norm, SimpleProjection, TopKPooling
from ..components.pooling import clip_k_to_shortest


class CLIPModel(nn.Module):
    def __init__(self, cfg, rank):
        super().__init__()
        self.cfg = cfg
        self.image_encoder = ImageEncoder(cfg)
        self.text_encoder = TextEncoder(cfg)
        self.random_seed = np.random.RandomState(seed=2021)
        heads = {"simple": SimpleProjection, "complex": ComplexProjection}
        if cfg.model.projection.name not in heads:
            raise NotImplementedError(cfg.model.projection.name)
        Head = heads[cfg.model.projection.name]
        self.image_projection = Head(cfg, embedding_dim=cfg.model.image_encoder.embedding_dim, projection_dim=cfg.model.projection.dim,
                                     trainable=cfg.model.projection.image_projector_trainable)
        self.text_projection = Head(cfg, embedding_dim=cfg.model.text_encoder.embedding_dim, projection_dim=cfg.model.projection.dim,
                                    trainable=cfg.model.projection.text_projector_trainable)
        if cfg.model.pool.name == "loda":
            self.text_pool = TopKPooling(cfg.model.pool.loda.text_k, dim=1)
            self.image_pool = TopKPooling(cfg.model.pool.loda.image_k, dim=1)
        elif cfg.model.pool.name == "avg":
            self.text_pool, self.image_pool = AvgPooling(), AvgPooling()
        else:
            self.text_pool, self.image_pool = nn.Identity(), nn.Identity()
        self.loss = LOSS.get(cfg.loss.name)(cfg, rank)
        self.global_reduce = cfg.loss.global_reduce
        self.text_target_token_idx = cfg.model.text_encoder.target_token_idx
        self._fused_heads = cfg.model.pool.name == "loda" and cfg.model.projection.name == "simple"

    # ---- image side ---------------------------------------------------------------------------------------------
    def forward_image_feature(self, image):
        """[B,3,H,W] -> patch tokens [B,N,D] (pool != identity) or the [cls] token [B,D] (clip.py:65-84)."""
        feats = self.image_encoder(image)
        if self.cfg.model.pool.name == "identity":
            return feats[:, 0] if feats.dim() == 3 else feats
        return feats[:, 1:] if feats.dim() == 3 else feats

    def _image_embeddings(self, image):
        """forward_image_project(forward_image_feature(image)) for the training / embedding path.  With the fused heads the tower's whole
        output goes to ProjectPoolFn, which leaves the [cls] token out of the pooling itself (skip=1) - no slice copy of the features, no
        zero-filled scatter of their gradient, no cast (simseg_amd/towers.py ProjectPoolFn).  Same values as the two calls."""
        if self._fused_heads and self.cfg.model.pool.name != "identity" and os.environ.get("SIMSEG_AMD_FUSED_IMAGE_HEAD", "1") != "0":
            feats = self.image_encoder(image)
            if feats.dim() == 3 and feats.is_cuda:
                return ProjectPoolFn.apply(feats, self.image_projection.linear.weight, self.image_pool.k, None, compute_dtype(), 1)
            return self.forward_image_project(feats[:, 1:] if feats.dim() == 3 else feats)
        return self.forward_image_project(self.forward_image_feature(image))

    def forward_image_project(self, image_features):
        """projection -> LoDA pool -> L2norm: [B,N,D] -> [B,P] (clip.py:87-93), one fused node."""
        if self._fused_heads and image_features.dim() == 3:
            return ProjectPoolFn.apply(image_features, self.image_projection.linear.weight, self.image_pool.k, None, compute_dtype())
        emb = self.image_pool(self.image_projection(image_features))
        return L2norm(emb, dim=-1) if self.cfg.model.projection.name == "simple" else emb

    # ---- text side ----------------------------------------------------------------------------------------------
    def forward_text_feature(self, input_ids, attention_mask):
        feats = self.text_encoder(input_ids=input_ids, attention_mask=attention_mask)
        if self.cfg.model.pool.name == "identity":
            return feats[:, self.text_target_token_idx, :]
        return feats[:, self.text_target_token_idx:, :]

    def forward_text_project(self, text_features, attention_mask):
        if self.cfg.model.pool.name == "identity":
            emb = self.text_pool(self.text_projection(text_features))
            return L2norm(emb, dim=-1) if self.cfg.model.projection.name == "simple" else emb
        mask = attention_mask[:, self.text_target_token_idx:] if self.text_target_token_idx else attention_mask
        if self._fused_heads:
            k = clip_k_to_shortest(self.text_pool.k, mask)
            return ProjectPoolFn.apply(text_features, self.text_projection.linear.weight, k, mask.contiguous().long(), compute_dtype())
        emb = self.text_pool(self.text_projection(text_features), mask)
        return L2norm(emb, dim=-1) if self.cfg.model.projection.name == "simple" else emb

    # ---- loss / dispatch ----------------------------------------------------------------------------------------
    def _prefetch(self, emb, embeddings):
        """Start the global-batch all-gather of one tower's embeddings (the exchange step of mml_loss.py:60-64) as soon as they
        exist, so that it runs under the other tower instead of in front of the loss."""
        if embeddings is False and self.global_reduce and getattr(self.loss, "group", None) is not None:
            prefetch_gather(emb, self.loss.group)

    def forward_loss(self, image_embeddings, text_embeddings, ignore_mask=None):
        if (self.global_reduce and ignore_mask is None and hasattr(self.loss, "both") and image_embeddings.is_cuda
                and os.environ.get("SIMSEG_AMD_FUSED_LOSS", "1") != "0"):
            # both directions as one node (same arithmetic and kernels as the two calls below: tests/test_gpu_model.py)
            loss, i2t_acc, t2i_acc = self.loss.both(image_embeddings, text_embeddings)
        elif self.global_reduce:
            i2t_loss, i2t_acc = self.loss(image_embeddings, text_embeddings, ignore_mask=ignore_mask)
            t2i_loss, t2i_acc = self.loss(text_embeddings, image_embeddings, ignore_mask=ignore_mask)
            loss = 0.5 * (i2t_loss + t2i_loss)
        else:
            loss, i2t_acc, t2i_acc = self.loss(image_embeddings, text_embeddings, ignore_mask=ignore_mask)
        return {f"{self.cfg.loss.name}_loss".lower(): loss}, i2t_acc, t2i_acc

    def forward(self, batch, embeddings=False):
        if embeddings == "image":
            return self.forward_image_feature(batch["image"])
        if embeddings == "text":
            return self.forward_text_feature(batch["input_ids"], batch["attention_mask"])
        image, ids, mask = batch["image"], batch["input_ids"], batch["attention_mask"]
        # optional, beyond the reference's batch keys: the captions' token counts as HOST numbers (what the tokenizer returned before the
        # host->device copy).  With them the text tower sizes its packed rows without reading anything back from the GPU.
        lengths = batch.get("caption_lengths") if isinstance(batch, dict) else None
        discard_prefetched(logger.warning)           # gathers of an earlier step that never reached the loss
        self.two_streams_used = bool(image.is_cuda and _two_streams_ok())      # (read by callers that launch per-tower work on the tower's stream)
        if self.two_streams_used:
            # The two towers are independent until the loss: the text tower runs on a second HIP stream so that its kernels
            # fill the CUs the image tower's kernels leave idle (partial last rounds of the 256-CU tile grids, memory-bound
            # LayerNorm / attention phases).  autograd replays each tower's backward on the stream its forward used.
            main = torch.cuda.current_stream()
            side = _side_stream(image.device)
            side.wait_stream(main)
            # (which tower is enqueued first makes no measurable difference: 126.2 vs 126.4 ms/step)
            img = self._image_embeddings(image)
            self._prefetch(img, embeddings)          # the image embeddings travel while the text tower is still computing
            with torch.cuda.stream(side), packed_text(lengths):
                txt = self.forward_text_project(self.forward_text_feature(ids, mask), mask)
                self._prefetch(txt, embeddings)
            main.wait_stream(side)
            txt.record_stream(main)
            # The caption tensors were allocated under the main stream but are read by side-stream kernels in the forward AND, through
            # the saved tensors of the text tower, in its backward.  Without this mark the caching allocator hands their memory back to
            # the main stream the moment the last Python / autograd reference drops, while those kernels may still be queued (a batch
            # that goes out of scope after the step: wrong word-embedding and first-layer gradients, found by the full-size parity test).
            for t in (ids, mask):
                if t.is_cuda:
                    t.record_stream(side)
        else:
            img = self._image_embeddings(image)
            self._prefetch(img, embeddings)
            with packed_text(lengths):      # only the masked pooling reads the text tower's output here: padded token rows are not computed
                txt = self.forward_text_project(self.forward_text_feature(ids, mask), mask)
        if embeddings == "all":
            return [img, txt]
        return self.forward_loss(img, txt, ignore_mask=None)


_SIDE = {}


def _two_streams_ok():
    """SIMSEG_AMD_TWO_STREAMS: 0 = never, 1 = always; default = unless gradients flow through torch DDP (its bucket hooks
    synchronise the all-reduce with ONE stream - the one that produced the last gradient of a bucket - which is not a
    guarantee this repo can test on a single-GPU box when gradients come from two streams)."""
    env = os.environ.get("SIMSEG_AMD_TWO_STREAMS")
    if env is not None:
        return env != "0"
    multi = torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1
    return not (multi and torch.is_grad_enabled())


def _side_stream(device):
    st = _SIDE.get(device)
    if st is None:
        st = _SIDE[device] = torch.cuda.Stream(device=device, priority=int(os.environ.get("SIMSEG_AMD_SIDE_PRIORITY", "0")))      # (A/B: -1 = the text tower's stream at high priority)
        # text-tower gradients are produced on the side stream and accumulated on the parameters' stream: intended
        # (autograd inserts the synchronisation), so its advisory warning is switched off
        f = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
        if f is not None:
            f(False)
    return st


class ImageEncoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.model_tag = cfg.model.image_encoder.tag
        self.pretrained = cfg.model.image_encoder.pretrained
        self.trainable = cfg.model.image_encoder.trainable
        kwargs = {}
        if "vit" in self.model_tag:
            kwargs["img_size"] = cfg.transforms.input_size      # clip.py:193-194
        else:
            kwargs["global_pool"] = ""
        self.model = BACKBONE.get(cfg.model.image_encoder.name)(cfg, **kwargs)
        for p in self.model.parameters():
            p.requires_grad = self.trainable

    def forward(self, x):
        return self.model(x)


class TextEncoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.model_tag = cfg.model.text_encoder.tag
        self.pretrained = cfg.model.text_encoder.pretrained
        self.trainable = cfg.model.text_encoder.trainable
        self.model = BACKBONE.get(cfg.model.text_encoder.name)(cfg)
        for p in self.model.parameters():
            p.requires_grad = self.trainable

    def forward(self, input_ids, attention_mask):
        return self.model(input_ids=input_ids, attention_mask=attention_mask).last_hidden_state


@PIPELINE.register_obj
def clip(cfg):
    return CLIPModel(cfg, ENV.rank)
