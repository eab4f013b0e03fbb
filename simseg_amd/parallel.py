"""Data-parallel gradient synchronisation that keeps the two-stream tower schedule.

torch DDP launches its bucket all-reduces from its own autograd hooks and synchronises each with ONE stream (the one that completed
the bucket); with the two towers' backward passes on two HIP streams that is one stream too few, so under DDP the towers run on
one stream (simseg/models/pipelines/clip.py).  `GradSync` is this package's own exchange:

  * every trainable parameter's gradient lives in ONE flat fp32 buffer (views handed to the fused AdamW as `.grad`);
  * the buffer is cut into buckets in backward order (last layers first).  A post-accumulate-grad hook per
    parameter copies the fresh gradient into its view and records an event on the stream that produced it; when the last
    gradient of a bucket has arrived, the bucket's all-reduce is enqueued on a dedicated communication stream behind exactly
    those events - RCCL rings over xGMI are per-link bound, so buckets are large (default 64 MiB) and few;
  * `finish()` (after `loss.backward()`) makes the compute stream wait for the collectives and averages.

So the exchange of the last layers' gradients runs under the backward of the earlier layers, on both tower streams, and the
reference's all-reduce semantics (mean over ranks, simseg/core/hooks/dist.py:48-54 / DDP) are kept.  `overlap=False` gives the plain
form: one flat all-reduce after the backward.

    sync = GradSync(model.parameters())
    loss.backward(); sync.finish(); optimizer.step()
"""
import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, params, group=None, average=True, overlap=True, bucket_mb=64):
        self.params = [p for p in params if p.requires_grad]
        self.group, self.average, self.overlap = group, average, overlap
        self.grad_scale = 1.0
        dev = self.params[0].device
        # backward order ~ reverse registration order (heads, text tower, image tower for the CLIP model)
        order = list(reversed(range(len(self.params))))
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        self.views = [None] * len(self.params)
        self.buckets = []                     # [start, end) element ranges of the flat buffer + member indices
        cap = int(bucket_mb) * (1 << 20) // 4
        o, cur = 0, None
        for i in order:
            n = self.params[i].numel()
            if cur is None or cur["n"] + n > cap:
                cur = {"lo": o, "n": 0, "members": []}
                self.buckets.append(cur)
            self.views[i] = self.flat[o:o + n].view_as(self.params[i])
            cur["members"].append(i); cur["n"] += n
            o += n
        self._bucket_of = {}
        for b, bk in enumerate(self.buckets):
            for i in bk["members"]:
                self._bucket_of[i] = b
        self._on = False
        self._comm = None
        self._handles = []
        if overlap:
            for i, p in enumerate(self.params):
                self._handles.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))
        self._reset()

    def _reset(self):
        self._pending = [len(bk["members"]) for bk in self.buckets]
        self._events = [[] for _ in self.buckets]
        self._works = []

    def _world(self):
        return dist.get_world_size(self.group) if (dist.is_available() and dist.is_initialized()) else 1

    def _reduce(self, b):
        bk = self.buckets[b]
        seg = self.flat[bk["lo"]:bk["lo"] + bk["n"]]
        if self.flat.is_cuda:
            if self._comm is None:
                self._comm = torch.cuda.Stream(device=self.flat.device)
            for ev, _ in self._events[b]:
                self._comm.wait_event(ev)
            with torch.cuda.stream(self._comm):
                self._works.append(dist.all_reduce(seg, group=self.group, async_op=True))
        else:
            self._works.append(dist.all_reduce(seg, group=self.group, async_op=True))

    def _gather_bucket(self, b):
        """Every gradient of bucket b has been produced: ONE multi-tensor copy moves them into their views of the flat buffer (a copy
        kernel per parameter - ~400 launches of a few microseconds in the backward's critical path - cost 1.9 ms per step at 512 pairs)
        and the parameters' .grad become those views.  Runs on the stream of the bucket's last gradient, behind the events of the
        members that another stream produced."""
        src, dst = [], []
        for i in self.buckets[b]["members"]:
            p, v = self.params[i], self.views[i]
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                src.append(p.grad); dst.append(v)
        if self.flat.is_cuda:
            cur = torch.cuda.current_stream()
            for ev, st in self._events[b]:
                if st != cur:
                    cur.wait_event(ev)
        if dst:
            torch._foreach_copy_(dst, src)
        for i in self.buckets[b]["members"]:
            self.params[i].grad = self.views[i]
        if self.flat.is_cuda:
            ev = torch.cuda.Event()
            ev.record()
            self._events[b] = [(ev, torch.cuda.current_stream())]

    def _make_hook(self, i):
        def hook(p):
            v = self.views[i]
            b = self._bucket_of[i]
            if self._pending[b] <= 0:
                # this bucket's all-reduce is already enqueued: a second backward before finish() (gradient accumulation) would add
                # into the flat views while that collective may still be in flight - partly reduced gradients and a data race
                raise RuntimeError("GradSync(overlap=True): a parameter's gradient arrived again after its bucket was reduced - exactly one "
                                   "backward per finish(); for gradient accumulation use GradSync(overlap=False) or accumulate inside one backward")
            if self.flat.is_cuda:
                ev = torch.cuda.Event()
                ev.record()                       # on the stream this gradient was produced on (main or the text tower's)
                self._events[b].append((ev, torch.cuda.current_stream()))
            self._pending[b] -= 1
            if self._pending[b] == 0:
                self._gather_bucket(b)
                if self._world() > 1:
                    self._reduce(b)
        return hook

    @torch.no_grad()
    def finish(self):
        """Call after backward: gradients that never arrived count as zero; the compute stream waits for every bucket; mean over ranks."""
        world = self._world()
        if not self.overlap:
            src, dst = [], []
            for p, v in zip(self.params, self.views):
                if p.grad is None:
                    v.zero_()
                elif p.grad.data_ptr() != v.data_ptr():
                    src.append(p.grad); dst.append(v)
            if dst:
                torch._foreach_copy_(dst, src)
            if world > 1:
                dist.all_reduce(self.flat, group=self.group)
        else:
            for b, left in enumerate(self._pending):
                if left:                           # a bucket with parameters that got no gradient this step (they count as zero)
                    self._gather_bucket(b)
                    if world > 1:
                        self._reduce(b)
            for w in self._works:
                w.wait()                           # the current stream waits for the collective (RCCL: no host block)
            if self.flat.is_cuda and self._comm is not None:
                torch.cuda.current_stream().wait_stream(self._comm)
        # average = "defer": the division by the world size is left to the optimizer kernel (AdamW.step(grad_scale=sync.grad_scale)) -
        # one pass over 0.8 GB of gradients less per step
        self.grad_scale = 1.0 / world if (world > 1 and self.average == "defer") else 1.0
        if world > 1 and self.average and self.average != "defer":
            self.flat.div_(world)
        for p, v in zip(self.params, self.views):
            p.grad = v
        self._reset()

    __call__ = finish
