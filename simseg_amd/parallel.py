"""Data-parallel gradient synchronisation without autograd hooks.

torch DDP launches its bucket all-reduces from AccumulateGrad hooks and synchronises each with the stream of the hook that
completed the bucket; with the two towers' backward passes on two HIP streams that is one stream too few, so DDP training runs
the towers on one stream (simseg/models/pipelines/clip.py).  `GradSync` keeps the two-stream schedule instead: after
`loss.backward()` has returned (autograd has joined its streams), every gradient is copied into ONE flat fp32 buffer, the buffer
is all-reduced over RCCL in a single collective (xGMI rings are per-link bound: one large message, not many small ones) and the
parameters' `.grad` are re-pointed at views of the reduced buffer - the fused AdamW reads them in place.

    sync = GradSync(model.parameters())
    loss.backward(); sync(); optimizer.step()

The gradient exchange is not overlapped with the backward; what is bought is the 6 % of the two-stream schedule."""
import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, params, group=None, average=True):
        self.params = [p for p in params if p.requires_grad]
        self.group, self.average = group, average
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        self.views, o = [], 0
        for p in self.params:
            self.views.append(self.flat[o:o + p.numel()].view_as(p))
            o += p.numel()

    @torch.no_grad()
    def __call__(self):
        on = dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1
        src, dst = [], []
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                src.append(p.grad); dst.append(v)
        if dst:
            torch._foreach_copy_(dst, src)
        if on:
            dist.all_reduce(self.flat, group=self.group)
            if self.average:
                self.flat.div_(dist.get_world_size(self.group))
        for p, v in zip(self.params, self.views):
            p.grad = v
