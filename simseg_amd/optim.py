"""AdamW on the fused HIP kernel (simseg_adamw_step).  Same update rule and hyper-parameters as the reference's
torch.optim.AdamW (configs/clip/simseg.vit-b.yaml:31-36); state is fp32 (m, v) per parameter."""
import torch

from . import ops


class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=1e-3):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._step = 0

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        self._step += 1
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["m"], st["v"] = torch.zeros_like(p), torch.zeros_like(p)
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                ops.adamw_step(p, g, st["m"], st["v"], None, group["lr"], group["betas"], group["eps"], group["weight_decay"],
                               self._step, grad_scale)
