"""Local pretrained tower weights.  The reference pulls timm / HF hub weights at build time (vit_builder.py:11,
huggingface_builder.py:10-11); there is no network here, so `pretrained: True` reads
$SIMSEG_PRETRAINED_DIR/<tag>.pth (a plain state dict in timm / HF naming).

Like `timm.create_model(tag, pretrained=True, num_classes=0, img_size=S)`, a ViT checkpoint whose position embedding was trained
on another patch grid (the shipped YAMLs pair `vit_*_patch16_224_in21k` with `input_size: 288`) has `pos_embed` resampled
bicubically to the model's grid, and classifier-head keys are dropped.  The reference always starts from pretrained towers, so a
missing file is an error unless SIMSEG_ALLOW_RANDOM_INIT=1 says the caller loads a full SimSeg checkpoint right afterwards (what
tools/seg_evaluation.py:225-233 and tools/retrieval_evaluation.py do) or really wants a from-scratch run."""
import os

import torch

from simseg.utils import logger
from simseg.utils.interpolate_pe import interpolate_pos_embed

_KNOWN_UNEXPECTED = ("position_ids",)       # a buffer of transformers 4.21.3 checkpoints, not a parameter here
_HEAD_PREFIXES = ("head.", "head_dist.", "pre_logits.", "fc_norm.", "cls.", "pooler.", "bert.pooler.")


def _adapt(sd, module):
    sd = {k: v for k, v in sd.items() if not k.startswith(_HEAD_PREFIXES)}
    sd = {(k[len("bert."):] if k.startswith("bert.") else k): v for k, v in sd.items()}       # BertForPreTraining-style prefixes
    # the original bert-base-uncased checkpoints name LayerNorm's parameters gamma / beta; HF's from_pretrained renames them on load
    # (LayerNorm modules only: timm's LayerScale parameters are called `ls1.gamma` / `ls2.gamma` and must keep their names)
    def _ln(k):
        if "LayerNorm." in k and k.endswith(".gamma"):
            return k[:-len(".gamma")] + ".weight"
        if "LayerNorm." in k and k.endswith(".beta"):
            return k[:-len(".beta")] + ".bias"
        return k
    sd = {_ln(k): v for k, v in sd.items()}
    pe = sd.get("pos_embed")
    if pe is not None and hasattr(module, "pos_embed") and hasattr(module, "patch_embed") and pe.shape != module.pos_embed.shape:
        sd["pos_embed"] = interpolate_pos_embed(pe.float(), module)
    return sd


def maybe_load_pretrained(module, tag):
    root = os.environ.get("SIMSEG_PRETRAINED_DIR")
    path = os.path.join(root, tag + ".pth") if root else None
    if path and os.path.exists(path):
        sd = _adapt(torch.load(path, map_location="cpu"), module)
        missing, unexpected = module.load_state_dict(sd, strict=False)
        unexpected = [k for k in unexpected if not k.endswith(_KNOWN_UNEXPECTED)]
        # `embeddings.position_ids` is a persistent BUFFER of this tower (as in transformers 4.21.3); checkpoints written by later
        # transformers versions and the original gamma / beta files do not carry it - its constructor value (arange) is the only one
        missing = [k for k in missing if not k.endswith(_KNOWN_UNEXPECTED)]
        if missing:
            # a file that only partly matches would leave those tensors at their random / identity init while reporting "pretrained"
            raise KeyError(f"pretrained weights for {tag!r} ({path}) do not cover {len(missing)} tensors of the tower, e.g. "
                           f"{sorted(missing)[:8]}; keys the file has that the tower does not: {sorted(unexpected)[:8]}")
        if unexpected:
            logger.warning(f"pretrained {tag}: {len(unexpected)} keys of {path} are not used: {sorted(unexpected)[:8]}")
        logger.info(f"loaded pretrained {tag} from {path} (every tensor of the tower covered)")
        return True
    if os.environ.get("SIMSEG_ALLOW_RANDOM_INIT", "0") not in ("", "0"):
        logger.warning(f"pretrained weights for {tag} are not available offline; keeping the random init (SIMSEG_ALLOW_RANDOM_INIT)")
        return False
    raise FileNotFoundError(
        f"`pretrained: True` for {tag!r} but no weights were found (looked for {path or '$SIMSEG_PRETRAINED_DIR/' + tag + '.pth'}). "
        "There is no network access to the timm / HuggingFace hubs: put a state dict in timm / HF naming there, pass "
        "`model.<tower>.pretrained=False`, or set SIMSEG_ALLOW_RANDOM_INIT=1 when a full SimSeg checkpoint is loaded afterwards.")
