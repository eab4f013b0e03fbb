"""One training iteration with the reference's semantics, as a compact driver over the HIP path:

  CLIPRunner.batch_processor   (simseg/tasks/clip/clip_runner.py:216-251)  autocast forward -> loss dict
  OptimizerHook                (simseg/core/hooks/optimizer.py:58-87)      stateless LR set, zero_grad, backward, clip, step
  ClipOptimizerHook            (simseg/tasks/clip/hooks/optimizer.py:18-36) one param group per parameter + regex rules
  lr lambdas                   (simseg/core/optimizer/lr_scheduler.py)      constant / warmup / linear / cosine (+ min_lr_scale)
  gen_checkpoint               (simseg/core/hooks/checkpoint.py:14-45)      {state_dict, optimizer, meta[, scaler]}

Differences that follow from the hardware path: 16-bit compute is bf16, so no loss scaling is needed (the `scaler` entry
of a checkpoint is written as an identity GradScaler state and ignored on load); the reference's per-step
torch.cuda.empty_cache() (clip_runner.py:248-249) is not reproduced."""
import math
import os
import re
import time

import torch

from .optim import AdamW, GradScaler


# ---- stateless learning-rate multipliers (functions of the global step only) -----------------------------------------
def lr_multiplier(name, step, num_warmup_steps=0, num_training_steps=1, num_cycles=0.5, min_lr_scale=0.0, **_):
    if name == "constant_schedule":
        return 1.0
    warm = step < num_warmup_steps
    if "warmup" in name and warm:
        return float(step) / float(max(1, num_warmup_steps))
    if name == "constant_schedule_with_warmup":
        return 1.0
    progress = float(step - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))
    if name == "linear_schedule_with_warmup":
        return max(0.0, 1.0 - progress)
    cos = 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress))
    if name == "cosine_schedule_with_warmup":
        return max(0.0, cos)
    if name == "cosine_schedule_with_warmup_min_lr_scale":          # lr_scheduler.py:192-221, the shipped YAMLs' schedule
        if not 0.0 <= min_lr_scale <= 1.0:
            raise AssertionError(f"min_lr_scale should be in [0, 1], but is {min_lr_scale}")
        return max(0.0, min_lr_scale + (1.0 - min_lr_scale) * cos)
    raise KeyError(f"unknown lr schedule {name!r}")


def param_groups(model, cfg):
    """One group per trainable parameter; cfg.optim.param_group_rules = {name: {regex, param: {...}}} override lr / wd."""
    base_lr, base_wd = cfg.optim.lr.init, cfg.optim.param["weight_decay"]
    groups = []
    for key, value in model.named_parameters():
        if not value.requires_grad:
            continue
        g = {"params": [value], "lr": base_lr, "weight_decay": base_wd}
        for rule in cfg.optim.param_group_rules.values():
            if re.search(rule["regex"], key):
                g.update(rule.get("param", {}))
        groups.append(g)
    return groups


class Trainer:
    def __init__(self, model, cfg, steps_per_epoch, net=None, amp_dtype=None):
        """model: the bare CLIPModel; net: what is called (model or its DDP wrapper).
        amp_dtype ("bf16" default / "fp16", or SIMSEG_AMD_AMP_DTYPE): the 16-bit type of the AMP mode cfg.dist.fp16 switches on.  "fp16" is
        the reference's own mode - torch.cuda.amp.autocast() + a live GradScaler (clip_runner.py:226-230, core/hooks/optimizer.py:73-82):
        scaled loss, unscale + overflow check + skipped steps, scale growth / back-off, the scaler's state in the checkpoint.  bf16 needs
        no loss scale (fp32's exponent range): the scaler then runs disabled, as an identity."""
        self.model, self.net, self.cfg = model, net or model, cfg
        amp = str(amp_dtype or os.environ.get("SIMSEG_AMD_AMP_DTYPE", "bf16")).lower().replace("torch.", "")
        if amp not in ("bf16", "bfloat16", "fp16", "float16", "half"):
            raise ValueError(f"amp_dtype {amp!r}: bf16 or fp16")
        self.amp_dtype = torch.float16 if amp in ("fp16", "float16", "half") else torch.bfloat16
        # (torch.amp.GradScaler with this package's overflow check: one read-only kernel; no host read in scaler.step either way)
        self.scaler = GradScaler("cuda", enabled=bool(cfg.dist.fp16) and self.amp_dtype == torch.float16)
        groups = param_groups(model, cfg)
        p = dict(cfg.optim.param)
        if cfg.optim.name.endswith("AdamW"):
            self.optimizer = AdamW(groups, lr=cfg.optim.lr.init, betas=tuple(p.get("betas", (0.9, 0.999))), eps=p.get("eps", 1e-8),
                                   weight_decay=p.get("weight_decay", 1e-2), half_dtype=self.amp_dtype)
        else:
            import importlib
            mod, _, cls = cfg.optim.name.rpartition(".")
            self.optimizer = getattr(importlib.import_module(mod or "torch.optim"), cls)(groups, lr=cfg.optim.lr.init, **p)
        self.base_lrs = [g["lr"] for g in self.optimizer.param_groups]
        total = steps_per_epoch * cfg.epoch
        warm = 0
        if cfg.optim.lr.warmup_proportion is not None:
            warm = int(total * cfg.optim.lr.warmup_proportion)
        if cfg.optim.lr.warmup_epoch is not None:
            warm = int(steps_per_epoch * cfg.optim.lr.warmup_epoch)
        self.sched = dict(name=cfg.optim.lr.name, num_warmup_steps=warm, num_training_steps=total, **dict(cfg.optim.lr.param))
        self.step, self.epoch, self.inner_step = 0, 0, 0

    def set_lrs(self, step):
        m = lr_multiplier(step=step, **self.sched)
        lrs = [b * m for b in self.base_lrs]
        for g, lr in zip(self.optimizer.param_groups, lrs):
            g["lr"] = lr
        return lrs

    def train_step(self, batch):
        lrs = self.set_lrs(self.step)
        self.optimizer.zero_grad(set_to_none=self.net is self.model)
        with torch.autocast("cuda", dtype=self.amp_dtype, enabled=bool(self.cfg.dist.fp16)):
            loss_dict, i2t_acc, t2i_acc = self.net(batch)
        loss = sum(loss_dict.values())
        self.scaler.scale(loss).backward()           # (disabled scaler: the loss itself)
        clip = dict(self.cfg.optim.grad_clip)
        if clip:                                     # the reference clips BEFORE scaler.step (on the scaled gradients): core/hooks/optimizer.py:82-86
            torch.nn.utils.clip_grad_norm_([p for p in self.model.parameters() if p.grad is not None], **clip)
        self.scaler.step(self.optimizer)             # unscale_, skip the step on inf / nan (disabled: optimizer.step())
        self.scaler.update()
        self.step += 1
        self.inner_step += 1
        return {"loss": loss.detach(), "i2t_acc": i2t_acc, "t2i_acc": t2i_acc, "lr": lrs[0]}

    # ---- checkpoints in the reference's layout ---------------------------------------------------------------------
    def checkpoint(self, end_of_epoch=False):
        meta = dict(time=time.asctime(), simseg_version="0.1.0+mi355x", torch_version=torch.__version__,
                    epoch=self.epoch + 1 if end_of_epoch else self.epoch, step=self.step, inner_step=0 if end_of_epoch else self.inner_step)
        return dict(state_dict=self.model.state_dict(), optimizer=self.optimizer.state_dict(), meta=meta,
                    scaler=self.scaler.state_dict())

    def load_checkpoint(self, state, load_optimizer=True):
        sd = state.get("state_dict") or state.get("model_state_dict") or state.get("model")      # tasks/clip/hooks/checkpoint.py:58-76
        sd = {k[len("module."):] if k.startswith("module.") else k: v for k, v in sd.items()}
        missing, unexpected = self.model.load_state_dict(sd, strict=False)
        if load_optimizer and "optimizer" in state:
            self.optimizer.load_state_dict(state["optimizer"])
            self.optimizer._plans.clear() if hasattr(self.optimizer, "_plans") else None
        if self.scaler.is_enabled() and state.get("scaler"):
            self.scaler.load_state_dict(state["scaler"])
        meta = state.get("meta", {})
        self.step, self.epoch, self.inner_step = meta.get("step", 0), meta.get("epoch", 0), meta.get("inner_step", 0)
        return missing, unexpected
