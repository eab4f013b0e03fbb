"""Communication side of the data-parallel training step (SURVEY.md 8d "xGMI link", 8e): a PREFLIGHT that runs every collective of the path
once with a value check before anything is timed, and the arithmetic of the bench line's `roofline.comm` block.

The path's collectives (reference: simseg/utils/dist.py:323-354 GatherLayer, core/hooks/dist.py:48-51 gradient all-reduce):
  C1  all-gather of the [Bl, 512] fp32 image and text embeddings (heads.ClipLossFn / GatherLayer forward), twice per step
  C2  reduce-scatter of the [W * Bl, 512] fp32 gradients of the gathered embeddings (their backward), twice per step
  C4  all-reduce of the fp32 gradient buckets (parallel.GradSync), ~780 MB per rank-step for ViT-B + BERT-base
  --  all-reduce MIN / MAX of two int64 (GradSync's check that every rank cut its buckets the same way), once

Runs on any backend (`nccl` = RCCL on the GPUs; `gloo` in the CPU tests) and on a process group of ONE rank (SIMSEG_FORCE_COLLECTIVES=1 on
a single-GPU box: the collectives are issued, the expected values are those of one rank)."""
import time

import torch
import torch.distributed as dist

XGMI_LINKS = 7
XGMI_LINK_GBPS = 153.0            # /opt/skills/guides/MI355X_MICROARCH.md: 7 links x ~153 GB/s per GPU, point to point
XGMI_PEAK_GBPS = XGMI_LINKS * XGMI_LINK_GBPS


class PreflightError(RuntimeError):
    pass


def _timed(fn, dev, iters):
    fn()
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) * 1e-3 / iters
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    return (time.perf_counter() - t0) / iters


def wire_bytes(kind, world, payload_bytes):
    """Bytes one rank SENDS (= receives) for one collective on `payload_bytes` of local data, ring / direct algorithms alike:
    all-gather of a local block: (W-1) blocks received; reduce-scatter of a W-block buffer: (W-1)/W of it; all-reduce: 2 (W-1)/W."""
    w = max(1, int(world))
    if kind == "all_gather":
        return (w - 1) * payload_bytes
    if kind == "reduce_scatter":
        return (w - 1) * payload_bytes // w
    if kind == "all_reduce":
        return 2 * (w - 1) * payload_bytes // w
    raise ValueError(kind)


def preflight(dev, pairs_per_rank, bucket_bytes=64 << 20, group=None, iters=5, ranks_info=None):
    """Every collective of the step once, values checked against what `world` ranks must produce; then `iters` timed repetitions each.
    Raises PreflightError (the same decision on every rank: the verdicts are all-reduced) when a value is wrong or - under `nccl` - two
    ranks report the same device.  Returns {collective: {payload_bytes, wire_bytes_per_rank, seconds, algbw_GBps, busbw_GBps,
    frac_of_xgmi_peak, ok}}."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    backend = dist.get_backend(group)
    problems = []
    if backend == "nccl" and ranks_info is not None:
        devs = [r.get("pci", r.get("device")) for r in ranks_info]
        if len(set(devs)) != len(devs):
            problems.append(f"two ranks report the same device under nccl: {devs}")
    tri = world * (world + 1) // 2
    out = {}

    def record(name, kind, payload, fn, ok):
        sec = _timed(fn, dev, iters)
        wb = wire_bytes(kind, world, payload)
        out[name] = {"collective": kind, "payload_bytes": int(payload), "wire_bytes_per_rank": int(wb), "seconds": sec,
                     "algbw_GBps": round(payload / sec / 1e9, 2), "busbw_GBps": round(wb / sec / 1e9, 2),
                     "frac_of_xgmi_peak": round(wb / sec / 1e9 / XGMI_PEAK_GBPS, 4), "ok": bool(ok)}
        if not ok:
            problems.append(f"{name}: wrong values on rank {rank}")

    # C1: all-gather of [Bl, 512] fp32
    loc = torch.full((pairs_per_rank, 512), float(rank + 1), device=dev, dtype=torch.float32)
    gat = torch.zeros(world * pairs_per_rank, 512, device=dev, dtype=torch.float32)
    dist.all_gather_into_tensor(gat, loc, group=group)
    want = torch.arange(1, world + 1, device=dev, dtype=torch.float32).repeat_interleave(pairs_per_rank)
    ok = bool(torch.equal(gat[:, 0], want) and torch.equal(gat[:, 511], want))
    record("C1_embedding_all_gather", "all_gather", loc.numel() * 4, lambda: dist.all_gather_into_tensor(gat, loc, group=group), ok)
    # C2: reduce-scatter of [W * Bl, 512] fp32: block r of every rank's buffer holds (rank + 1) * (r + 1)
    blk = torch.arange(1, world + 1, device=dev, dtype=torch.float32).repeat_interleave(pairs_per_rank)
    buf = (blk[:, None] * float(rank + 1)).expand(-1, 512).contiguous()
    own = torch.zeros(pairs_per_rank, 512, device=dev, dtype=torch.float32)
    dist.reduce_scatter_tensor(own, buf, op=dist.ReduceOp.SUM, group=group)
    ok = bool(torch.all(own == float(tri * (rank + 1))))
    record("C2_embedding_grad_reduce_scatter", "reduce_scatter", buf.numel() * 4, lambda: dist.reduce_scatter_tensor(own, buf, op=dist.ReduceOp.SUM, group=group), ok)
    # C4: all-reduce of one gradient bucket (fp32)
    n = max(1, int(bucket_bytes) // 4)
    g = torch.full((n,), float(rank + 1), device=dev, dtype=torch.float32)
    dist.all_reduce(g, group=group)
    ok = bool(g[0] == float(tri) and g[n - 1] == float(tri) and g[n // 2] == float(tri))
    g.fill_(1.0)
    record("C4_gradient_bucket_all_reduce", "all_reduce", n * 4, lambda: dist.all_reduce(g, group=group), ok)
    # GradSync's bucket-cut check
    sig = torch.tensor([rank + 3, 7], device=dev, dtype=torch.int64)
    lo, hi = sig.clone(), sig.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    ok = lo.tolist() == [3, 7] and hi.tolist() == [world + 2, 7]
    out["bucket_cut_check_int64_min_max"] = {"collective": "all_reduce", "payload_bytes": 16, "ok": bool(ok)}
    if not ok:
        problems.append(f"int64 MIN / MAX all-reduce: wrong values on rank {rank}")
    # the verdict is the same on every rank
    bad = torch.tensor([len(problems)], device=dev, dtype=torch.int64)
    dist.all_reduce(bad, op=dist.ReduceOp.MAX, group=group)
    if int(bad) > 0:
        raise PreflightError("; ".join(problems) if problems else "another rank's preflight failed")
    return out


def step_traffic(world, pairs_per_rank, gradient_bytes, embed_dim=512):
    """Bytes per rank and step of each collective class of the contrastive step: {class: (count, payload, wire bytes)}."""
    emb = pairs_per_rank * embed_dim * 4
    return {"C1_embedding_all_gather": {"per_step": 2, "payload_bytes_each": emb, "wire_bytes_per_rank_step": 2 * wire_bytes("all_gather", world, emb)},
            "C2_embedding_grad_reduce_scatter": {"per_step": 2, "payload_bytes_each": world * emb,
                                                 "wire_bytes_per_rank_step": 2 * wire_bytes("reduce_scatter", world, world * emb)},
            "C4_gradient_all_reduce": {"per_step": None, "payload_bytes_each": None, "payload_bytes_per_step": int(gradient_bytes),
                                       "wire_bytes_per_rank_step": wire_bytes("all_reduce", world, int(gradient_bytes))}}


def comm_roofline(world, pairs_per_rank, gradient_bytes, n_buckets, probe, ms_step, ms_step_no_comm, process_group=None):
    """The bench line's `roofline.comm`: bytes per rank-step of C1 / C2 / C4, what the isolated collectives reach (the preflight's timings)
    against the xGMI peak, how long they would take back to back at those rates, and how much of that the step did not hide."""
    tr = step_traffic(world, pairs_per_rank, gradient_bytes)
    tr["C4_gradient_all_reduce"]["per_step"] = n_buckets
    iso_ms = 0.0
    for name, ent in tr.items():
        pr = (probe or {}).get({"C4_gradient_all_reduce": "C4_gradient_bucket_all_reduce"}.get(name, name))
        if pr and pr.get("busbw_GBps"):
            ent["isolated_busbw_GBps"] = pr["busbw_GBps"]
            ent["isolated_frac_of_xgmi_peak"] = pr["frac_of_xgmi_peak"]
            ent["ms_per_step_at_isolated_rate"] = round(1e3 * ent["wire_bytes_per_rank_step"] / (pr["busbw_GBps"] * 1e9), 3) if pr["busbw_GBps"] > 0 else None
            iso_ms += ent["ms_per_step_at_isolated_rate"] or 0.0
    total = sum(e["wire_bytes_per_rank_step"] for e in tr.values())
    exposed = None if ms_step_no_comm is None else round(ms_step - ms_step_no_comm, 3)
    return {"bound": "xgmi", "peak": XGMI_PEAK_GBPS, "unit": "GB/s", "peak_is": f"{XGMI_LINKS} xGMI links x {XGMI_LINK_GBPS:.0f} GB/s per GPU, point to point",
            "world_size": world, "wire_bytes_per_rank_step": int(total), "per_class": tr,
            "ms_per_step_if_nothing_overlapped": round(iso_ms, 3),
            "ms_per_step": round(ms_step, 3), "ms_per_step_with_local_stand_ins": None if ms_step_no_comm is None else round(ms_step_no_comm, 3),
            "exposed_communication_ms_per_step": exposed,
            "exposed_is": "step(N) minus the same step on the same ranks with every collective replaced by a local stand-in of the same shapes "
                          "(heads.LOCAL_STANDIN, GradSync.skip_collectives): buffers, hooks, events and the communication stream stay",
            "achieved": None if not exposed or exposed <= 0 else round(total / (exposed * 1e-3) / 1e9, 1),
            "achieved_is": "wire bytes per rank-step / exposed time: a LOWER bound of the rate the collectives ran at (hidden time not counted)",
            "frac": None if not exposed or exposed <= 0 else round(total / (exposed * 1e-3) / 1e9 / XGMI_PEAK_GBPS, 4),
            "process_group": process_group}
