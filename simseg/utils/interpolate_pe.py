"""Bicubic resize of a checkpoint's ViT position embedding to the model's patch grid
(mirror of simseg/utils/interpolate_pe.py:4-26; load-time, not on the hot path -> torch)."""
import torch.nn.functional as F

__all__ = ["interpolate_pos_embed"]


def interpolate_pos_embed(pos_embed_checkpoint, visual_encoder):
    dim = pos_embed_checkpoint.shape[-1]
    n_new = visual_encoder.patch_embed.num_patches
    n_extra = visual_encoder.pos_embed.shape[-2] - n_new            # [cls] (and dist) tokens stay as they are
    side_old = int((pos_embed_checkpoint.shape[-2] - n_extra) ** 0.5)
    side_new = int(n_new ** 0.5)
    if side_old == side_new:
        return pos_embed_checkpoint
    extra, grid = pos_embed_checkpoint[:, :n_extra], pos_embed_checkpoint[:, n_extra:]
    grid = grid.reshape(-1, side_old, side_old, dim).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, size=(side_new, side_new), mode="bicubic", align_corners=False)
    grid = grid.permute(0, 2, 3, 1).flatten(1, 2)
    print("reshape position embedding from %d to %d" % (side_old ** 2, side_new ** 2))
    return __import__("torch").cat((extra, grid), dim=1)
