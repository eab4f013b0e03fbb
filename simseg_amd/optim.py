"""AdamW on the fused HIP kernel.  Same update rule and hyper-parameters as the reference's torch.optim.AdamW
(configs/clip/simseg.vit-b.yaml:31-36); state (m, v) is fp32.

ONE kernel launch updates every parameter tensor of every param group (the reference's ClipOptimizerHook builds one group per
parameter, tasks/clip/hooks/optimizer.py:18-36, so "one launch per group" would be ~400 launches): a device table of
{p, g, m, v, p16, lr, weight_decay} rows is refreshed from pinned host memory each step (gradient tensors are re-allocated by
autograd, learning rates follow the schedule) without a host-device synchronisation.  The pinned staging buffers form a ring, each
guarded by an event recorded after its upload, so the host never rewrites a buffer whose asynchronous copy has not executed yet.
p16 is the bf16 compute copy of the parameter, written by the same kernel and handed to the towers (towers.register_w16), so a
training step launches no weight-cast kernels.

Checkpoints use torch.optim.AdamW's state layout ({step, exp_avg, exp_avg_sq} per parameter), so the reference's optimizer
checkpoints (core/hooks/checkpoint.py:14-45) load here and ours load there; the bias-correction step counter travels with them."""
import contextlib

import numpy as np
import torch

from .lib import call, note_half, ptr, stream
from .towers import drop_split_copy, register_w16

CHUNK = 1 << 16
RING = 4


class AdamW(torch.optim.Optimizer):
    # torch.amp.GradScaler's contract for optimizers that handle the loss scale themselves (grad_scaler.py: `_step_supports_amp_scaling`):
    # scaler.step(optimizer) then sets optimizer.grad_scale / optimizer.found_inf (device tensors) and calls step() WITHOUT reading the
    # overflow flag on the host; the kernel unscales the gradients and skips the update on the device (simseg_adamw_multi_step_amp).
    _step_supports_amp_scaling = True

    def __init__(self, params, lr=1e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=1e-3, half_dtype=torch.bfloat16):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        if half_dtype not in (torch.bfloat16, torch.float16):
            raise TypeError("half_dtype: torch.bfloat16 (the default) or torch.float16 (the reference's AMP type, the benchmark headline since round 5)")
        self.half_dtype = half_dtype      # type of the 16-bit compute copies the kernel writes (what the towers' GEMMs read next step)
        self._step = 0                    # step() calls without a GradScaler (every one of them updates)
        self._step_dev = None             # AMP: float32 [2] on the device, the count of steps actually TAKEN (skipped ones do not count)
        self._amp_calls = 0               #      which slot is current
        self._plans = {}
        self._prepared = None
        self._streams = {}                # id(parameter) -> stream its update is launched on (set_param_streams); default: the current stream

    def set_param_streams(self, mapping):
        """mapping: {parameter: torch.cuda.Stream or None}.  The update of those parameters is launched on that stream instead of the current
        one - for a tower whose forward AND backward run on a stream of their own (the text tower of CLIPModel's two-stream schedule) it
        then starts when that tower's backward ends instead of waiting behind the other tower's: ~0.5 ms of a 512-pair step.  The caller owns
        the ordering: every later reader of these parameters (and of their 16-bit copies) must run on that stream or wait for it, and
        their gradients must be complete on it (single-process training; with a gradient exchange leave them on the current stream)."""
        for p_, st in mapping.items():
            if st is None:
                self._streams.pop(id(p_), None)
            else:
                self._streams[id(p_)] = st
        self._plans.clear()
        self._prepared = None

    # ---- launch plan: everything about a set of tensors that does not change from step to step ------------------------------
    def _plan(self, key, params):
        ids = tuple(id(p) for p in params)
        plan = self._plans.get(key)
        if plan is not None and plan["ids"] == ids:
            return plan
        dev = params[0].device
        total = sum(p.numel() for p in params)
        m = torch.zeros(total, device=dev, dtype=torch.float32)
        v = torch.zeros(total, device=dev, dtype=torch.float32)
        p16 = torch.empty(total, device=dev, dtype=self.half_dtype)
        o = 0
        for p in params:
            st = self.state[p]
            n = p.numel()
            if "m" in st:                     # keep moments across a re-plan / a loaded checkpoint
                m[o:o + n].copy_(st["m"].reshape(-1)); v[o:o + n].copy_(st["v"].reshape(-1))
            st["m"], st["v"] = m[o:o + n].view_as(p), v[o:o + n].view_as(p)
            o += n
        # bf16 copies: matrices first, in parameter order - the weights of modules that share an input (BERT's query / key / value)
        # then sit back to back and the towers use them as one [3D, D] operand without a concatenation (towers._wt_stacked)
        o = 0
        for p in sorted(params, key=lambda q: q.dim() < 2):
            n = p.numel()
            self.state[p]["p16"] = p16[o:o + n].view_as(p)
            o += n
        tid, coff = [], []
        for t, p in enumerate(params):
            for c in range(0, p.numel(), CHUNK):
                tid.append(t); coff.append(c)
        static = np.zeros((len(params), 6), dtype=np.int64)
        for t, p in enumerate(params):
            st = self.state[p]
            static[t, 0], static[t, 2], static[t, 3], static[t, 4] = p.data_ptr(), st["m"].data_ptr(), st["v"].data_ptr(), st["p16"].data_ptr()
        plan = dict(ids=ids, m=m, v=v, p16=p16, static=static,
                    sizes=torch.tensor([p.numel() for p in params], dtype=torch.int64, device=dev),
                    tid=torch.tensor(tid, dtype=torch.int32, device=dev), coff=torch.tensor(coff, dtype=torch.int64, device=dev),
                    ring=[torch.empty(len(params), 6, dtype=torch.int64).pin_memory() for _ in range(RING)],
                    events=[None] * RING, slot=0,
                    table=torch.empty(len(params), 6, dtype=torch.int64, device=dev), n_chunks=len(tid))
        self._plans[key] = plan
        return plan

    def _prepare(self, on_own=True):
        """Per-step tables of every bucket (gradient pointers, learning rates) uploaded; reused by the call that follows immediately
        (found_inf_check() then step() inside one scaler.step)."""
        grads_now = tuple(id(p.grad) for g in self.param_groups for p in g["params"])
        if self._prepared is not None and self._prepared[0] == grads_now:
            return self._prepared[1]
        buckets = {}
        for group in self.param_groups:
            key0 = (float(group["betas"][0]), float(group["betas"][1]), float(group["eps"]))
            for p in group["params"]:
                if p.grad is not None:
                    st = self._streams.get(id(p))
                    key = key0 + ((st.cuda_stream,) if st is not None else ())
                    buckets.setdefault(key, []).append((p, float(group["lr"]), float(group["weight_decay"])))
        ready = []
        for key, items in buckets.items():
            params = [it[0] for it in items]
            if any(not p.is_contiguous() or p.dtype != torch.float32 for p in params):
                raise TypeError("simseg_amd AdamW expects contiguous fp32 master parameters")
            plan = self._plan(key, params)
            if any(p.data_ptr() != a for p, a in zip(params, plan["static"][:, 0])):       # storage swapped (.to(), load with assign)
                self._plans.pop(key)
                plan = self._plan(key, params)
            grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in params]
            plan["stream"] = self._streams.get(id(params[0])) if (len(key) > 3 and on_own) else None      # where this step's upload + launch go
            slot = plan["slot"]
            plan["slot"] = (slot + 1) % RING
            if plan["events"][slot] is not None:
                plan["events"][slot].synchronize()          # the upload that last used this pinned buffer has executed (RING steps ago)
            host = plan["ring"][slot].numpy()
            host[:] = plan["static"]
            host[:, 1] = [g.data_ptr() for g in grads]
            hyper = np.empty((len(params), 2), dtype=np.float32)
            hyper[:, 0] = [it[1] for it in items]
            hyper[:, 1] = [it[2] for it in items]
            host[:, 5] = hyper.view(np.int64)[:, 0]
            with (torch.cuda.stream(plan["stream"]) if plan["stream"] is not None else contextlib.nullcontext()):
                plan["table"].copy_(plan["ring"][slot], non_blocking=True)      # (on the stream the kernel is launched on)
                ev = torch.cuda.Event()
                ev.record()
            plan["events"][slot] = ev
            plan["keepalive"] = grads        # the kernels read them asynchronously
            ready.append((key, plan, params))
        self._prepared = (grads_now, ready)
        return ready

    @torch.no_grad()
    def found_inf_check(self, found_inf):
        """found_inf (float32 scalar tensor on the device) = 1 if any gradient holds an inf / nan: GradScaler's overflow check as one
        read-only launch per bucket over this optimizer's tensor table (simseg_amd.optim.GradScaler calls it instead of torch's
        read-modify-write pass over every gradient tensor)."""
        for key, plan, params in self._prepare(on_own=False):
            call("simseg_grads_nonfinite", ptr(plan["table"]), ptr(plan["sizes"]), ptr(plan["tid"]), ptr(plan["coff"]), plan["n_chunks"], CHUNK,
                 ptr(found_inf), stream())
        return found_inf

    def steps_taken(self):
        """Updates actually applied (a host read of the device counter when a GradScaler drives this optimizer: skipped steps do not count)."""
        if self._step_dev is not None:
            return int(round(float(self._step_dev[self._amp_calls & 1])))
        return self._step

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0, param_streams=True):
        # torch.amp.GradScaler.step() sets these two attributes around the call (and deletes them afterwards)
        loss_scale, found_inf = getattr(self, "grad_scale", None), getattr(self, "found_inf", None)
        amp = loss_scale is not None or found_inf is not None
        # param_streams=False (or a GradScaler-driven step, whose overflow flag is produced on the current stream): every launch on the
        # current stream, whatever set_param_streams said - the caller's backward did not run on those streams this time
        on_own = param_streams and not amp
        ready = self._prepare(on_own)
        self._prepared = None
        if amp:
            if self._step_dev is None:       # the device counter takes over from the host one
                dev = ready[0][1]["table"].device if ready else torch.device("cuda")
                self._step_dev = torch.full((2,), float(self._step), device=dev, dtype=torch.float32)
                self._amp_calls = 0
            for t in (loss_scale, found_inf):
                if t is not None and (t.dtype != torch.float32 or not t.is_cuda):
                    raise TypeError("grad_scale / found_inf: float32 tensors on the device (torch.amp.GradScaler's)")
        else:
            if self._step_dev is not None:   # back from a GradScaler-driven phase: one host read of how many of its steps were taken
                self._step = self.steps_taken()
                self._step_dev = None
            self._step += 1
        for key, plan, params in ready:
          own = plan.get("stream")
          with (torch.cuda.stream(own) if own is not None else contextlib.nullcontext()):
              note_half(self.half_dtype)        # (the 16-bit copies are addressed through the table: tell the binding which flavour they are)
              if amp:
                  cur = self._amp_calls & 1
                  call("simseg_adamw_multi_step_amp", ptr(plan["table"]), ptr(plan["sizes"]), ptr(plan["tid"]), ptr(plan["coff"]), plan["n_chunks"],
                       CHUNK, key[0], key[1], key[2], float(grad_scale), ptr(loss_scale), ptr(found_inf), ptr(self._step_dev[cur:cur + 1]),
                       ptr(self._step_dev[1 - cur:2 - cur]), stream())
              else:
                  call("simseg_adamw_multi_step", ptr(plan["table"]), ptr(plan["sizes"]), ptr(plan["tid"]), ptr(plan["coff"]), plan["n_chunks"],
                       CHUNK, key[0], key[1], key[2], self._step, float(grad_scale), stream())
              for p in params:                 # same stream as the next forward: the copies are current when it runs
                  register_w16(p, self.state[p]["p16"])
                  drop_split_copy(p)           # the exact-mode split-bf16 copy of the OLD value (raw-pointer update: _version did not move)
        if amp and ready:
            self._amp_calls += 1             # (several buckets: every launch of this call read the same slot and wrote the other; a call
                                             #  with no gradient at all launched nothing - the other slot was not written, so do not flip)

    # ---- checkpoints in torch.optim.AdamW's layout ---------------------------------------------------------------------
    def state_dict(self):
        sd = super().state_dict()
        steps = self.steps_taken()
        out = {}
        for idx, st in sd["state"].items():
            if "m" in st:
                out[idx] = {"step": torch.tensor(float(steps)), "exp_avg": st["m"].clone(), "exp_avg_sq": st["v"].clone()}
        sd["state"] = out
        for g in sd["param_groups"]:            # keys torch.optim.AdamW.load_state_dict expects to find
            g.setdefault("amsgrad", False)
        return sd

    def load_state_dict(self, state_dict):
        sd = dict(state_dict)
        steps, conv = [], {}
        for idx, st in sd["state"].items():
            if "exp_avg" in st:
                conv[idx] = {"m": st["exp_avg"], "v": st["exp_avg_sq"]}
                steps.append(int(float(st["step"])))
            elif "m" in st:                     # round-1 checkpoints of this package
                conv[idx] = {"m": st["m"], "v": st["v"]}
        sd["state"] = conv
        if "_step" in state_dict:
            steps.append(int(state_dict["_step"]))
        super().load_state_dict(sd)
        self._plans.clear()                     # moments are re-packed (and the bf16 copies rewritten) by the next step
        for st in self.state.values():
            st["m"], st["v"] = st["m"].float().contiguous(), st["v"].float().contiguous()
            st.pop("p16", None)
        self._step = max(steps) if steps else 0
        self._step_dev, self._amp_calls, self._prepared = None, 0, None


class GradScaler(torch.amp.GradScaler):
    """torch.amp.GradScaler (same state, same state_dict, same scale / step / update calls as the reference makes: clip_runner.py:226-230,
    core/hooks/optimizer.py:73-82) whose overflow check of a simseg_amd AdamW is that optimizer's one read-only kernel instead of torch's
    read-modify-write pass over ~400 gradient tensors.  With either scaler the step has no host read: the skip decision stays on the device
    (AdamW._step_supports_amp_scaling)."""

    def __init__(self, device="cuda", **kw):
        super().__init__(device, **kw)

    def _check_inf_per_device(self, optimizer):
        if not isinstance(optimizer, AdamW):
            return super()._check_inf_per_device(optimizer)
        _scale, _ = self._check_scale_growth_tracker("_check_inf_per_device")
        found_inf = torch.zeros((), dtype=torch.float32, device=_scale.device)
        optimizer.found_inf_check(found_inf)
        self._per_optimizer_states[id(optimizer)]["found_inf_per_device"] = {_scale.device: found_inf}
        return self._per_optimizer_states[id(optimizer)]["found_inf_per_device"]
