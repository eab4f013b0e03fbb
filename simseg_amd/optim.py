"""AdamW on the fused HIP kernels.  Same update rule and hyper-parameters as the reference's torch.optim.AdamW
(configs/clip/simseg.vit-b.yaml:31-36); state (m, v) is fp32.  All parameter tensors of a param group are updated by ONE
kernel launch (simseg_adamw_multi_step): a device table of {p, g, m, v, p16} pointers is refreshed from a pinned host buffer each
step (gradient tensors are re-allocated by autograd), without a host-device synchronisation.  p16 is the bf16 compute copy of
the parameter, written by the same kernel and handed to the towers (towers.register_w16), so a training step launches no
weight-cast kernels."""
import numpy as np
import torch

from . import ops
from .lib import call, ptr, stream
from .towers import register_w16

CHUNK = 1 << 16


class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=1e-3):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._step = 0
        self._plans = {}

    def _plan(self, gi, params):
        """Static part of a group's launch: state buffers, sizes, chunk map (rebuilt when the set of tensors changes)."""
        key = tuple(id(p) for p in params)
        plan = self._plans.get(gi)
        if plan is not None and plan["key"] == key:
            return plan
        dev = params[0].device
        total = sum(p.numel() for p in params)
        m = torch.zeros(total, device=dev, dtype=torch.float32)
        v = torch.zeros(total, device=dev, dtype=torch.float32)
        p16 = torch.empty(total, device=dev, dtype=torch.bfloat16)
        old = self._plans.get(gi)
        offs, o = [], 0
        for p in params:
            offs.append(o)
            st = self.state[p]
            if "m" in st:                     # keep moments across a re-plan
                m[o:o + p.numel()].copy_(st["m"].reshape(-1)); v[o:o + p.numel()].copy_(st["v"].reshape(-1))
            st["m"], st["v"] = m[o:o + p.numel()].view_as(p), v[o:o + p.numel()].view_as(p)
            st["p16"] = p16[o:o + p.numel()].view_as(p)
            o += p.numel()
        tid, coff = [], []
        for t, p in enumerate(params):
            for c in range(0, p.numel(), CHUNK):
                tid.append(t); coff.append(c)
        plan = dict(key=key, m=m, v=v, p16=p16,
                    sizes=torch.tensor([p.numel() for p in params], dtype=torch.int64, device=dev),
                    tid=torch.tensor(tid, dtype=torch.int32, device=dev), coff=torch.tensor(coff, dtype=torch.int64, device=dev),
                    host=torch.empty(len(params), 5, dtype=torch.int64).pin_memory(),
                    table=torch.empty(len(params), 5, dtype=torch.int64, device=dev), n_chunks=len(tid))
        self._plans[gi] = plan
        return plan

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        self._step += 1
        for gi, group in enumerate(self.param_groups):
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            if any(not p.is_contiguous() or p.dtype != torch.float32 for p in params):
                raise TypeError("simseg_amd AdamW expects contiguous fp32 master parameters")
            plan = self._plan(gi, params)
            grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in params]
            host = plan["host"].numpy()
            for t, (p, g) in enumerate(zip(params, grads)):
                st = self.state[p]
                host[t, 0], host[t, 1], host[t, 2], host[t, 3] = p.data_ptr(), g.data_ptr(), st["m"].data_ptr(), st["v"].data_ptr()
                host[t, 4] = st["p16"].data_ptr()
            plan["table"].copy_(plan["host"], non_blocking=True)
            wd = plan.get("wd")
            if wd is None or plan.get("wd_val") != group["weight_decay"]:
                wd = plan["wd"] = torch.full((len(params),), float(group["weight_decay"]), device=params[0].device)
                plan["wd_val"] = group["weight_decay"]
            call("simseg_adamw_multi_step", ptr(plan["table"]), ptr(plan["sizes"]), ptr(plan["tid"]), ptr(plan["coff"]), plan["n_chunks"],
                 CHUNK, float(group["lr"]), float(group["betas"][0]), float(group["betas"][1]), float(group["eps"]), ptr(wd),
                 self._step, float(grad_scale), stream())
            plan["keepalive"] = grads        # the kernel reads them asynchronously
            for p in params:                 # same stream as the next forward: the copies are current when it runs
                register_w16(p, self.state[p]["p16"])
