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
    sync.begin(); loss = model(batch); loss.backward(); sync.finish(); optimizer.step()

`begin()` (optional, before the forward pass) zero-fills the flat buffer and arms the ZERO-COPY path: the towers' backward functions
(simseg_amd/towers.py `_grad_target`) then accumulate the large weight gradients - the split-K GEMMs add into a zero-filled output
anyway - straight into their views of the flat buffer, autograd adopts those views as `.grad`, and the bucket hand-over has nothing to
copy for them (the copies of a 64 MiB bucket were 14 x ~140 us of a 512-pair step, profiles/r4_gradsync_zero_copy.txt).  Without
`begin()` every gradient is copied into its view as before.
"""
import os

import torch
import torch.distributed as dist

_EVENTS = os.environ.get("SIMSEG_GRADSYNC_EVENTS", "needed")      # "all": an event per parameter, as before round 4 (A/B runs)


def force_collectives():
    """SIMSEG_FORCE_COLLECTIVES=1: a process group of ONE rank still issues every collective of the data path (all-reduce of the gradient
    buckets, the int64 MIN / MAX check of the bucket cut; heads.py: embedding all-gather / reduce-scatter) instead of taking the one-rank
    shortcuts.  The arithmetic is the same; what it buys is running the N > 1 code path - communication stream, events, Work handles,
    record_stream hand-overs - on real RCCL on a box with a single GPU (RCCL refuses two ranks per device, but serves a one-rank group)."""
    return os.environ.get("SIMSEG_FORCE_COLLECTIVES", "0") == "1" and dist.is_available() and dist.is_initialized()


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
        self.skip_collectives = False         # benchmarks only (bench.py's communication-free leg): the whole machinery, no all-reduce
        self._comm = None
        self._handles = []
        self._armed = False
        self._handed = [False] * len(self.params)
        self._split_done = False
        # Which gradients need an event of their own.  A bucket is handed over on the stream that produced its LAST gradient; members
        # produced on that same stream are ordered by the stream itself, and with the towers on two streams that is all but a handful per
        # step (the buckets at a tower boundary).  An event record per parameter - ~400 barrier packets with release fences in the two compute
        # streams - measured 2.2 ms of a 86.5 ms step (profiles/r4_gradsync_events_ab.txt), so only the members that were produced on ANOTHER
        # stream than their bucket's hand-over in the previous step record one (first step: all); a member that turns out to need one
        # without having it gets a late event on its stream at the hand-over (correct; waits for more than it has to, once).
        self._need_ev = [True] * len(self.params)
        self.events_last = None
        self._nev = 0
        self.copied_last = None               # gradients that had to be copied into their views in the last step (tests / bench)
        self._copied = 0
        for i, p in enumerate(self.params):
            p._simseg_grad_target = self._make_target(i)
        if overlap:
            for i, p in enumerate(self.params):
                self._handles.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))
        self._reset()

    def _make_target(self, i):
        def target():
            """A FRESH view of parameter i's slice of the flat buffer (autograd adopts a gradient tensor nobody else references), or None
            when this step was not armed with begin()."""
            if not self._armed:
                return None
            # Handed out ONCE per step and only while the parameter has no gradient tensor yet.  Otherwise autograd's AccumulateGrad would
            # run `p.grad += view` with both aliasing the same memory (a stale flat view kept by zero_grad(set_to_none=False), or a second
            # Function application of the same parameter in one backward): 2x gradients, silently.  Returning None makes the backward
            # function allocate its own gradient tensor, which autograd then accumulates - and the bucket hand-over copies - as usual.
            if self._handed[i] or self.params[i].grad is not None:
                return None
            self._handed[i] = True
            v = self.views[i]
            return v.view(v.shape)
        return target

    @torch.no_grad()
    def begin(self):
        """Before the forward pass of a step: zero the flat buffer (on the current stream - the tower streams fork from it later) and let the
        backward functions write the large weight gradients straight into it.  Precondition for the zero-copy hand-over of a parameter:
        its .grad is None when its backward runs (zero_grad(set_to_none=True)) and it feeds one Function application per backward; a
        parameter that does not meet it simply takes the copying path (_make_target)."""
        self.flat.zero_()
        self._armed = True
        self._handed = [False] * len(self.params)
        self._copied = 0

    def close(self):
        for h in self._handles:
            h.remove()
        self._handles = []
        for p in self.params:
            if hasattr(p, "_simseg_grad_target"):
                del p._simseg_grad_target

    def _split_by_stream(self):
        """After the first step: a bucket whose gradients were produced on more than one stream (the towers run on two; a bucket at a tower
        boundary holds the last layers of one and the first of the other) is cut into its runs of same-stream members.  Its hand-over
        otherwise makes the stream that completes it wait for the OTHER tower's stream - the two towers serialise there, 1.5-2 ms of an
        86.5 ms step with the default 64 MiB buckets (profiles/r4_gradsync_events_ab.txt) - and with stream-pure buckets no hand-over waits for
        anything but its own stream.  The flat layout does not change: a run is a contiguous sub-range."""
        self._split_done = True
        out = []
        for bk in self.buckets:
            lo, run = bk["lo"], None
            for i in bk["members"]:
                st = self._prod[i]
                if run is None or (st is not None and run["st"] is not None and st != run["st"]):
                    run = {"lo": lo, "n": 0, "members": [], "st": st}
                    out.append(run)
                elif run["st"] is None:
                    run["st"] = st
                run["members"].append(i); run["n"] += self.params[i].numel()
                lo += self.params[i].numel()
        if self._world() > 1 or force_collectives():       # every rank must cut the same way (the collectives are per bucket): same code, same streams - checked once
            sig = torch.tensor([len(out), sum((k + 1) * r["lo"] for k, r in enumerate(out)) % (1 << 40)], device=self.flat.device, dtype=torch.int64)
            lo_, hi_ = sig.clone(), sig.clone()
            dist.all_reduce(lo_, op=dist.ReduceOp.MIN, group=self.group)
            dist.all_reduce(hi_, op=dist.ReduceOp.MAX, group=self.group)
            if not torch.equal(lo_, hi_):
                return                  # (ranks disagree: keep the common, uncut buckets)
        if len(out) != len(self.buckets):
            self.buckets = [{"lo": r["lo"], "n": r["n"], "members": r["members"]} for r in out]
            self._bucket_of = {i: b for b, bk in enumerate(self.buckets) for i in bk["members"]}

    def _reset(self):
        self._pending = [len(bk["members"]) for bk in self.buckets]
        self._events = [[] for _ in self.buckets]
        self._works = []
        self._prod = [None] * len(self.params)      # the stream each gradient was produced on in this step
        self._has_ev = [False] * len(self.params)

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
        self._src_stream = {}
        for i in self.buckets[b]["members"]:
            p, v = self.params[i], self.views[i]
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                src.append(p.grad); dst.append(v)
                self._src_stream[id(p.grad)] = self._prod[i]      # (None = not seen by a hook this step: treated as foreign)
        self._copied += len(dst)
        if self.flat.is_cuda:
            cur = torch.cuda.current_stream()
            for ev, st in self._events[b]:
                if st != cur:
                    cur.wait_event(ev)
            late = []
            for i in self.buckets[b]["members"]:
                st = self._prod[i]
                self._need_ev[i] = st is not None and st != cur
                if self._need_ev[i] and not self._has_ev[i] and st not in late:
                    late.append(st)
            for st in late:                       # (a member the previous step's pattern did not predict)
                ev = torch.cuda.Event()
                ev.record(st)
                cur.wait_event(ev)
                self._nev += 1
        if dst:
            torch._foreach_copy_(dst, src)
            if self.flat.is_cuda:
                # a source gradient produced on ANOTHER stream lives in that stream's allocator pool; `p.grad = view` below drops its last
                # reference, and without this the block could be handed back to its producer (still running its backward) and overwritten
                # before the copy enqueued on `cur` has executed
                cur = torch.cuda.current_stream()
                for g_ in src:
                    if self._src_stream.get(id(g_)) != cur:
                        g_.record_stream(cur)
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
                st = torch.cuda.current_stream()  # the stream this gradient was produced on (main or the text tower's)
                self._prod[i] = st
                if self._need_ev[i] or _EVENTS == "all":
                    ev = torch.cuda.Event()
                    ev.record()
                    self._events[b].append((ev, st))
                    self._has_ev[i] = True
                    self._nev += 1
            self._pending[b] -= 1
            if self._pending[b] == 0:
                self._gather_bucket(b)
                if (self._world() > 1 or force_collectives()) and not self.skip_collectives:
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
            self._copied += len(dst)
            if dst:
                torch._foreach_copy_(dst, src)
            if (world > 1 or force_collectives()) and not self.skip_collectives:
                dist.all_reduce(self.flat, group=self.group)
        else:
            for b, left in enumerate(self._pending):
                if left:                           # a bucket with parameters that got no gradient this step (they count as zero)
                    self._gather_bucket(b)
                    if (world > 1 or force_collectives()) and not self.skip_collectives:
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
        self._armed = False
        self.copied_last, self._copied = self._copied, 0
        self.events_last, self._nev = self._nev, 0
        if self.overlap and self.flat.is_cuda and not self._split_done:
            self._split_by_stream()
        self._reset()

    __call__ = finish
