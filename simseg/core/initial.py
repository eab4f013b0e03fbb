"""Process / device initialisation the entry scripts call (mirror of simseg/core/initial.py:37-75).
One process per GPU; backend 'nccl' is RCCL over xGMI on ROCm.  On a box without GPUs (CPU tests) it falls back to
gloo so that the collective plumbing can be exercised."""
import os
import random

import numpy as np
import torch
import torch.distributed as dist

from simseg.utils import ENV, logger

__all__ = ["init_device"]


def init_device(cfg):
    seed = cfg.seed if cfg.seed is not None else 0
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    ENV.cfg = cfg
    ENV.dist_mode = cfg.dist.name
    has_gpu = torch.cuda.is_available()
    if has_gpu:
        torch.cuda.set_device(ENV.local_rank)
        torch.cuda.manual_seed_all(seed)
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # RCCL ("nccl") on GPUs as in the reference; SIMSEG_DIST_BACKEND exists for bring-up (e.g. gloo with several ranks
        # sharing one GPU, which RCCL refuses)
        backend = os.environ.get("SIMSEG_DIST_BACKEND") or ("nccl" if has_gpu else "gloo")
        dist.init_process_group(backend=backend, init_method="env://")
    ENV.rank, ENV.size = dist.get_rank(), dist.get_world_size()
    if ENV.size > 1:
        # The persistent 256x256 GEMM launches one workgroup per CU that owns it for the whole launch (512 threads x 256 VGPRs, 144 KiB of
        # LDS: nothing co-resides).  With collectives in flight - RCCL's all-reduce / all-gather kernels are a few dozen workgroups that need
        # CUs of their own - the launch leaves 16 of the 256 CUs free (read once, at the library's first GEMM; SIMSEG_GEMM_PP2_RESERVE
        # overrides).  Measured on one GPU: reserves of 0 / 8 / 32 CUs are within 0.4 ms of an 88 ms step.  Not measurable here with RCCL
        # itself (one GPU per box; RCCL refuses two ranks per device).
        os.environ.setdefault("SIMSEG_GEMM_PP2_RESERVE", "16")
    ENV.device = torch.device("cuda", ENV.local_rank) if has_gpu else torch.device("cpu")
    for name in ("batch_size", "batch_size_val"):
        bs = cfg.data.get(name)
        if isinstance(bs, int) and bs % ENV.size != 0 and bs >= ENV.size:
            raise AssertionError(f"data.{name}={bs} must be divisible by the world size {ENV.size}")
    logger.info(f"init_device: rank {ENV.rank}/{ENV.size} on {ENV.device}", root_only=False)
