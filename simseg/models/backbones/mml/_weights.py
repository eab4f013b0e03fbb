"""Optional local pretrained weights.  The reference pulls timm / HF hub weights (vit_builder.py:11,
huggingface_builder.py:10-11); there is no network here, so `pretrained: True` looks for
$SIMSEG_PRETRAINED_DIR/<tag>.pth (a plain state dict in timm / HF naming) and otherwise keeps the random init --
the eval tools load a full SimSeg checkpoint right after building the model (tools/seg_evaluation.py:225-233)."""
import os

import torch

from simseg.utils import logger


def maybe_load_pretrained(module, tag):
    root = os.environ.get("SIMSEG_PRETRAINED_DIR")
    path = os.path.join(root, tag + ".pth") if root else None
    if path and os.path.exists(path):
        missing, unexpected = module.load_state_dict(torch.load(path, map_location="cpu"), strict=False)
        logger.info(f"loaded pretrained {tag} from {path} (missing {len(missing)}, unexpected {len(unexpected)})")
    else:
        logger.warning(f"pretrained weights for {tag} are not available offline; keeping the random init "
                       f"(set SIMSEG_PRETRAINED_DIR or load a SimSeg checkpoint)")
