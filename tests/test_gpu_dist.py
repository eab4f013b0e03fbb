"""Two data-parallel ranks on MI355X hardware: the exchange step of the contrastive loss (embedding all-gather forward,
reduce-scatter backward: simseg/utils/dist.py:323-354) with the HIP loss / towers in between, against what the REFERENCE produced
in a 2-process run (tests/golden/clip_train_ws2.npz).

The GPU box has ONE GPU and RCCL refuses two ranks on one device, so the two processes share cuda:0 and talk over gloo (device
tensors are staged by the backend); everything between the collectives is the product path.  `nccl` is tried first so that the
same test exercises RCCL wherever two devices are visible."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import GOLD, REPO

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, backend, q):
    import sys
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank if torch.cuda.device_count() >= world else 0), SIMSEG_AMD_COMPUTE="bf16")
    dev = torch.device("cuda", int(os.environ["LOCAL_RANK"]))
    torch.cuda.set_device(dev)
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        from simseg.utils import ENV
        ENV.rank, ENV.size, ENV.local_rank = rank, world, dev.index
        from simseg.models.criteria.losses.mml_loss import NCE
        from test_gpu_model import _build
        g = np.load(os.path.join(GOLD, f"clip_train_ws{world}.npz"))

        def golden(name):
            return np.load(os.path.join(GOLD, name + ".npz"))

        res = {}
        # (1) the loss alone, exact fp32: HIP NCEFn + GatherLayer on given embeddings, ignore mask on, temperature gradient
        m = _build(golden).eval()                   # global_reduce=True, gather_backward=True (YAML defaults, as the fixture)
        nce = m.loss
        assert isinstance(nce, NCE) and nce.group is not None and nce.rank == rank
        f1 = torch.from_numpy(g[f"r{rank}.nce_f1"]).to(dev).requires_grad_(True)
        f2 = torch.from_numpy(g[f"r{rank}.nce_f2"]).to(dev).requires_grad_(True)
        ign = torch.from_numpy(g[f"r{rank}.nce_ign"]).to(dev)
        loss, acc = nce(f1, f2, ignore_mask=ign)
        loss.backward()
        res["nce"] = dict(loss=loss.item(), acc=acc.item(), g1=f1.grad.cpu().numpy(), g2=f2.grad.cpu().numpy(),
                          gt=nce.temperature.grad.item())
        nce.temperature.grad = None
        # (2) the whole training forward/backward of this rank's batch (bf16 towers, prefetched gathers)
        for two in ("0", "1"):
            os.environ["SIMSEG_AMD_TWO_STREAMS"] = two
            m.zero_grad(set_to_none=True)
            batch = {k: torch.from_numpy(g[f"r{rank}.{k}"]).to(dev) for k in ("image", "input_ids", "attention_mask")}
            loss_dict, a1, a2 = m(batch)
            loss_dict["nce_loss"].backward()
            torch.cuda.synchronize()
            params = dict(m.named_parameters())
            res[f"step{two}"] = dict(loss=loss_dict["nce_loss"].item(), a1=a1.item(), a2=a2.item(),
                                     grads={k[len(f"r{rank}.grad."):]: params[k[len(f"r{rank}.grad."):]].grad.float().cpu().numpy()
                                            for k in g.files if k.startswith(f"r{rank}.grad.")})
        from simseg_amd import heads
        res["prefetch_left"] = len(heads._PREFETCH)
        q.put((rank, res))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _cos(a, b):
    a, b = a.astype(np.float64).ravel(), b.astype(np.float64).ravel()
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))


@pytest.mark.parametrize("world", [2])
def test_two_ranks_match_reference_two_process_run(world):
    backend = "nccl" if torch.cuda.device_count() >= world else "gloo"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, backend, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    g = np.load(os.path.join(GOLD, f"clip_train_ws{world}.npz"))
    for r in range(world):
        o = out[r]["nce"]
        np.testing.assert_allclose(o["loss"], float(g[f"r{r}.nce_loss"]), rtol=2e-5)
        np.testing.assert_allclose(o["acc"], float(g[f"r{r}.nce_acc"]), atol=1e-6)
        np.testing.assert_allclose(o["g1"], g[f"r{r}.nce_g1"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(o["g2"], g[f"r{r}.nce_g2"], rtol=1e-4, atol=1e-6)      # summed over both ranks' losses (reduce-scatter)
        np.testing.assert_allclose(o["gt"], float(g[f"r{r}.nce_gt"]), rtol=1e-4)
        assert out[r]["prefetch_left"] == 0                                                # every prefetched gather was consumed
        for two in ("0", "1"):
            s = out[r][f"step{two}"]
            want = float(g[f"r{r}.loss"])
            assert abs(s["loss"] - want) < 2e-2 * abs(want), (r, two, s["loss"], want)
            assert abs(s["a1"] - float(g[f"r{r}.i2t_acc"])) < 1e-6 and abs(s["a2"] - float(g[f"r{r}.t2i_acc"])) < 1e-6
            bad = []
            for name, ours in s["grads"].items():
                ref = g[f"r{r}.grad.{name}"]
                c = _cos(ours, ref)
                ratio = float(np.linalg.norm(ours) / (np.linalg.norm(ref) + 1e-30))
                print(f"rank {r} two_streams={two} {name}: cos {c:.5f} norm ratio {ratio:.4f}")
                if not (c > 0.98 and 0.9 < ratio < 1.1):       # tiny bf16 towers, 3 pairs per rank: top-k / argmax flips under bf16 noise
                                                                # (an exchange bug is O(1): the exact-fp32 loss part above is the tight check)
                    bad.append((r, two, name, c, ratio))
            assert not bad, bad


def _worker_groups(rank, world, port, backend, group_size, fixture, q):
    """Four ranks, the loss gathered over sub-groups (cfg.loss.group_size, mml_loss.py:24-27): the product path end to end - this repo's
    generate_local_groups, NCE module, GatherLayer over the SUB-group and the HIP loss kernels - in exact fp32 against the reference's
    4-process run, then one whole forward/backward of the rank's batch."""
    import sys
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank if torch.cuda.device_count() >= world else 0), SIMSEG_AMD_COMPUTE="fp32")
    dev = torch.device("cuda", int(os.environ["LOCAL_RANK"]))
    torch.cuda.set_device(dev)
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        from simseg.utils import ENV
        ENV.rank, ENV.size, ENV.local_rank = rank, world, dev.index
        from test_gpu_model import _build
        g = np.load(os.path.join(GOLD, fixture + ".npz"))

        def golden(name):
            return np.load(os.path.join(GOLD, name + ".npz"))

        m = _build(golden, extra=[f"loss.group_size={group_size}"]).eval()
        nce = m.loss
        res = {"group": (dist.get_world_size(nce.group), nce.rank)}
        f1 = torch.from_numpy(g[f"r{rank}.nce_f1"]).to(dev).requires_grad_(True)
        f2 = torch.from_numpy(g[f"r{rank}.nce_f2"]).to(dev).requires_grad_(True)
        ign = torch.from_numpy(g[f"r{rank}.nce_ign"]).to(dev)
        loss, acc = nce(f1, f2, ignore_mask=ign)
        loss.backward()
        res["nce"] = dict(loss=loss.item(), acc=acc.item(), g1=f1.grad.cpu().numpy(), g2=f2.grad.cpu().numpy(), gt=nce.temperature.grad.item())
        m.zero_grad(set_to_none=True)
        batch = {k: torch.from_numpy(g[f"r{rank}.{k}"]).to(dev) for k in ("image", "input_ids", "attention_mask")}
        loss_dict, a1, a2 = m(batch)                   # exact fp32 towers + prefetched sub-group gathers
        loss_dict["nce_loss"].backward()
        torch.cuda.synchronize()
        params = dict(m.named_parameters())
        grads = {}
        for k in g.files:
            if k.startswith(f"r{rank}.grad."):
                name = k[len(f"r{rank}.grad."):]
                v = params[name].grad.float().cpu().numpy()
                grads[name] = v[:g[k].shape[0]] if v.ndim == 2 and v.shape != g[k].shape else v
        res["step"] = dict(loss=loss_dict["nce_loss"].item(), a1=a1.item(), a2=a2.item(), grads=grads)
        q.put((rank, res))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("group_size,fixture", [(2, "clip_train_ws4g2"), (4, "clip_train_ws4")])
def test_four_ranks_with_loss_sub_groups_match_reference(group_size, fixture):
    world = 4
    backend = "nccl" if torch.cuda.device_count() >= world else "gloo"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_groups, args=(r, world, port, backend, group_size, fixture, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=900) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    g = np.load(os.path.join(GOLD, fixture + ".npz"))
    for r in range(world):
        assert out[r]["group"] == (group_size, r % group_size)
        o = out[r]["nce"]
        np.testing.assert_allclose(o["loss"], float(g[f"r{r}.nce_loss"]), rtol=2e-5)
        np.testing.assert_allclose(o["acc"], float(g[f"r{r}.nce_acc"]), atol=1e-6)
        np.testing.assert_allclose(o["g1"], g[f"r{r}.nce_g1"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(o["g2"], g[f"r{r}.nce_g2"], rtol=1e-4, atol=1e-6)      # summed over the sub-group's losses (reduce-scatter)
        np.testing.assert_allclose(o["gt"], float(g[f"r{r}.nce_gt"]), rtol=1e-4)
        s = out[r]["step"]
        np.testing.assert_allclose(s["loss"], float(g[f"r{r}.loss"]), rtol=1e-4)
        assert abs(s["a1"] - float(g[f"r{r}.i2t_acc"])) < 1e-6 and abs(s["a2"] - float(g[f"r{r}.t2i_acc"])) < 1e-6
        for name, ours in s["grads"].items():           # exact fp32 kernels: every stored gradient against the REFERENCE's own
            ref = g[f"r{r}.grad.{name}"]
            err = float(np.abs(ours - ref).max()) / (float(np.abs(ref).max()) + 1e-30)
            assert err < 2e-4, (r, name, err)


def _worker_sync_and_retrieval(rank, world, port, backend, q):
    """(a) GradSync(overlap=True) with the towers on two streams == torch DDP's averaged gradients; (b) the multi-rank retrieval
    evaluation == the single-process one."""
    import sys
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank if torch.cuda.device_count() >= world else 0), SIMSEG_AMD_COMPUTE="fp32")
    dev = torch.device("cuda", int(os.environ["LOCAL_RANK"]))
    torch.cuda.set_device(dev)
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        from simseg.utils import ENV
        ENV.rank, ENV.size, ENV.local_rank = rank, world, dev.index
        from simseg_amd.parallel import GradSync
        from test_gpu_model import _build
        g = np.load(os.path.join(GOLD, f"clip_train_ws{world}.npz"))

        def golden(name):
            return np.load(os.path.join(GOLD, name + ".npz"))

        batch = {k: torch.from_numpy(g[f"r{rank}.{k}"]).to(dev) for k in ("image", "input_ids", "attention_mask")}
        res = {}
        # torch DDP (towers on one stream under it: simseg/models/pipelines/clip.py:_two_streams_ok)
        os.environ.pop("SIMSEG_AMD_TWO_STREAMS", None)
        m1 = _build(golden).eval()                      # eval(): BERT dropout off, so both runs see the same function
        ddp = torch.nn.parallel.DistributedDataParallel(m1, device_ids=[dev.index], gradient_as_bucket_view=True)
        ddp(batch)[0]["nce_loss"].backward()
        torch.cuda.synchronize()
        want = {n: p.grad.detach().float().cpu().numpy() for n, p in m1.named_parameters() if p.grad is not None}
        del ddp
        # this package's exchange, hooks + communication stream, several small buckets, the text tower on its own stream
        os.environ["SIMSEG_AMD_TWO_STREAMS"] = "1"
        m2 = _build(golden).eval()
        sync = GradSync(m2.parameters(), overlap=True, bucket_mb=0)        # bucket_mb=0: one bucket per parameter - the most hooks / events
        for _ in range(2):                                                    # second step: the hooks re-arm after finish()
            for p in m2.parameters():
                p.grad = None
            m2(batch)[0]["nce_loss"].backward()
            sync.finish()
        torch.cuda.synchronize()
        got = {n: p.grad.detach().float().cpu().numpy() for n, p in m2.named_parameters() if p.grad is not None}
        res["grads"] = (want, got, len(sync.buckets))
        # one large bucket set (the default 64 MiB) as well
        m3 = _build(golden).eval()
        sync3 = GradSync(m3.parameters(), overlap=True)
        m3(batch)[0]["nce_loss"].backward()
        sync3.finish()
        torch.cuda.synchronize()
        res["grads_default_buckets"] = {n: p.grad.detach().float().cpu().numpy() for n, p in m3.named_parameters() if p.grad is not None}
        # ... and armed with begin(): the blocks' weight-gradient GEMMs and the word-embedding scatter accumulate straight into the exchange
        # buffer (towers._grad_target); those gradients are adopted by autograd and never copied
        copied_plain = sync3.copied_last
        for p in m3.parameters():
            p.grad = None
        sync3.begin()
        m3(batch)[0]["nce_loss"].backward()
        sync3.finish()
        torch.cuda.synchronize()
        res["grads_zero_copy"] = ({n: p.grad.detach().float().cpu().numpy() for n, p in m3.named_parameters() if p.grad is not None},
                                  copied_plain, sync3.copied_last, len(sync3.params))
        os.environ.pop("SIMSEG_AMD_TWO_STREAMS", None)

        # (b) retrieval: 60 images x 5 captions, rows dealt to the ranks unevenly (rank 0: 170 rows, rank 1: 130)
        from simseg_amd.retrieval import evaluate_sharded, retrieval_metrics
        gen = torch.Generator().manual_seed(5)
        n_img, cap, P = 60, 5, 64
        base = torch.nn.functional.normalize(torch.randn(n_img, P, generator=gen), dim=-1)
        rows = n_img * cap
        full = {"image_embeddings": base.repeat_interleave(cap, 0),
                "text_embeddings": torch.nn.functional.normalize(base.repeat_interleave(cap, 0) + 0.35 * torch.randn(rows, P, generator=gen), dim=-1),
                "image_id": torch.arange(rows) // cap, "caption_id": torch.arange(rows)}
        lo, hi = (0, 170) if rank == 0 else (170, rows)
        shard = {k: v[lo:hi].to(dev) for k, v in full.items()}
        res["retr_multi"] = evaluate_sharded(shard, "coco")
        if rank == 0:
            res["retr_single"] = retrieval_metrics({k: v.to(dev) for k, v in full.items()}, "coco")
        q.put((rank, res))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_gradsync_overlap_equals_ddp_and_sharded_retrieval_equals_single(world):
    """the RCCL-side paths that had only run on gloo/CPU - GradSync's hooks, events and communication stream with
    the HIP towers on two streams against torch DDP's gradients, and the multi-rank retrieval evaluation (a14) against the
    single-process recalls."""
    backend = "nccl" if torch.cuda.device_count() >= world else "gloo"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_sync_and_retrieval, args=(r, world, port, backend, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for r in range(world):
        want, got, nb = out[r]["grads"]
        assert nb > 20 and set(want) == set(got)
        for name in want:
            scale = float(np.abs(want[name]).max()) + 1e-12
            # exact-fp32 kernels on both sides; the differences are the order of the two ranks' sum and atomics inside kernels
            np.testing.assert_allclose(got[name], want[name], rtol=0, atol=2e-5 * scale + 1e-9, err_msg=f"rank {r} {name}")
            np.testing.assert_allclose(out[r]["grads_default_buckets"][name], want[name], rtol=0, atol=2e-5 * scale + 1e-9, err_msg=f"rank {r} {name} (64 MiB buckets)")
        zc, copied_plain, copied_armed, nparams = out[r]["grads_zero_copy"]
        for name in want:
            scale = float(np.abs(want[name]).max()) + 1e-12
            np.testing.assert_allclose(zc[name], want[name], rtol=0, atol=2e-5 * scale + 1e-9, err_msg=f"rank {r} {name} (zero-copy exchange)")
        # every gradient copied without begin(); armed, the block weight matrices (7 per layer pair) and the word embeddings are not
        assert copied_plain == nparams and copied_armed < copied_plain - 6, (copied_plain, copied_armed, nparams)
    for name in out[0]["grads"][1]:              # averaged gradients are the same on every rank
        np.testing.assert_array_equal(out[0]["grads"][1][name], out[1]["grads"][1][name])
    assert out[1]["retr_multi"] is None
    multi, single = out[0]["retr_multi"], out[0]["retr_single"]
    assert multi == single, (multi, single)      # same dot products, same integer ranks: bit-equal
    assert 0.0 < multi["coco_I2T-R@1"] < 100.0 and len(multi) == 7


def _worker_ws1(rank, world, port, backend, q):
    import sys
    sys.path.insert(0, REPO)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    torch.cuda.set_device(0)
    dist.init_process_group(backend, rank=0, world_size=1)
    try:
        from simseg_amd.heads import ClipLossFn, NCEFn, all_gather_rows, GatherLayer
        g = torch.Generator().manual_seed(0)
        res = {}
        for gb in (False, True):
            img = torch.nn.functional.normalize(torch.randn(48, 64, generator=g), dim=1).cuda().requires_grad_(True)
            txt = torch.nn.functional.normalize(torch.randn(48, 64, generator=g), dim=1).cuda().requires_grad_(True)
            temp = torch.tensor(0.07, device="cuda", requires_grad=True)
            loss, _, _ = ClipLossFn.apply(img, txt, temp, dist.group.WORLD, 0, 0.0, gb)
            loss.backward()
            fused = (img.grad.clone(), txt.grad.clone(), temp.grad.clone())
            img.grad = txt.grad = temp.grad = None
            gather = (lambda t: GatherLayer.apply(t, dist.group.WORLD, 0)) if gb else (lambda t: all_gather_rows(t, dist.group.WORLD))
            l1, _ = NCEFn.apply(img, gather(txt), temp, None, None, 0, 0.0)
            l2, _ = NCEFn.apply(txt, gather(img), temp, None, None, 0, 0.0)
            (0.5 * (l1 + l2)).backward()
            res[gb] = [float((a - b).abs().max() / b.abs().max()) for a, b in zip(fused, (img.grad, txt.grad, temp.grad))] + [float(loss - 0.5 * (l1 + l2))]
        q.put(res)
    finally:
        dist.destroy_process_group()


def test_fused_loss_head_equals_unfused_in_a_one_rank_group():
    """A REAL process group of one rank: with gather_backward=False the gathered role carries no gradient (all_gather_rows detaches, as the
    reference's all_gather_group does, utils/dist.py:65-74) - the fused head ClipLossFn once let it through in exactly this configuration
    (up to 2x embedding gradients against NCE.forward's path); with gather_backward=True both roles carry it.  Fused == unfused either way."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_ws1, args=(0, 1, _free_port(), "gloo", q))
    p.start()
    res = q.get(timeout=300)
    p.join(60)
    for gb, errs in res.items():
        assert max(errs[:3]) < 1e-5 and abs(errs[3]) < 1e-6, (gb, errs)


def test_bench_self_launches_two_ranks_on_rccl():
    """`python bench.py --gpus 2` with no launcher around it: bench.py re-executes itself under torch.distributed.run, one rank per GPU, and
    the process group is RCCL (`nccl`) on two different devices.  Needs two visible devices (the 1-GPU test box skips it; the driver's
    8-GPU node runs it)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("one visible device: RCCL refuses two ranks on it (the gloo two-rank tests above cover the exchange arithmetic)")
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--pairs-per-gpu", "64",
                          "--no-seg", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    pg = line["config"]["process_group"]
    assert line["n_gpus"] == 2 and pg["backend"] == "nccl" and pg["world_size"] == 2 and "warning" not in pg, pg
    assert len({r["device"] for r in pg["ranks"]}) == 2
    assert line["config"]["global_batch"] == 128 and line["value"] > 0
    _check_comm_block(line, world=2)


def _check_comm_block(line, world):
    """roofline.comm of an N-rank (or forced one-rank) bench line: bytes per rank-step of C1 / C2 / C4, the preflight's isolated rates against
    the xGMI peak, the communication-free leg and what it exposes."""
    comm = line["roofline"]["comm"]
    assert comm is not None and comm["bound"] == "xgmi" and comm["peak"] == 7 * 153.0 and comm["world_size"] == world
    pre = comm["preflight"]
    assert all(v["ok"] for v in pre.values()) and {"C1_embedding_all_gather", "C2_embedding_grad_reduce_scatter", "C4_gradient_bucket_all_reduce"} <= set(pre)
    pc = comm["per_class"]
    bl = line["config"]["pairs_per_gpu"]
    assert pc["C1_embedding_all_gather"]["wire_bytes_per_rank_step"] == 2 * (world - 1) * bl * 512 * 4
    assert pc["C2_embedding_grad_reduce_scatter"]["wire_bytes_per_rank_step"] == 2 * (world - 1) * bl * 512 * 4
    gb = pc["C4_gradient_all_reduce"]["payload_bytes_per_step"]
    assert 7.5e8 < gb < 8.2e8 and pc["C4_gradient_all_reduce"]["wire_bytes_per_rank_step"] == 2 * (world - 1) * gb // world       # ViT-B + BERT-base + heads in fp32
    assert pc["C4_gradient_all_reduce"]["per_step"] == line["config"]["gradient_sync_detail"]["buckets"]
    assert comm["ms_per_step_with_local_stand_ins"] is not None and comm["ms_per_step"] == line["ms_per_step"]
    assert abs(comm["exposed_communication_ms_per_step"] - (comm["ms_per_step"] - comm["ms_per_step_with_local_stand_ins"])) < 2e-3
    assert comm["process_group"]["world_size"] == world


def test_bench_comm_roofline_on_one_rank_rccl_group():
    """The N > 1 form of bench.py on the one-GPU box: a one-rank `nccl` (RCCL) group with SIMSEG_FORCE_COLLECTIVES=1 runs the preflight (every
    collective once, values checked), the headline with every collective issued, the communication-free leg, and reports roofline.comm."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", SIMSEG_FORCE_COLLECTIVES="1", SIMSEG_BENCH_FORCE_SYNC="1", SIMSEG_BENCH_OTHER_DTYPE_LEG="0",
               SIMSEG_BENCH_GELU16_LEG="0", SIMSEG_AMD_PACKED_TEXT="1")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "3", "--warmup", "2", "--pairs-per-gpu", "64", "--no-seg",
                          "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = out.stdout.strip().splitlines()
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["config"]["process_group"]["backend"] == "nccl" and line["n_gpus"] == 1
    _check_comm_block(line, world=1)
    assert "communication preflight ok" in out.stderr


def test_bench_two_ranks_gloo_bringup_reports_comm_roofline():
    """Two ranks sharing the one GPU over gloo (`SIMSEG_DIST_BACKEND=gloo SIMSEG_BENCH_DEVICE=0 python bench.py --gpus 2`, self-launched): the
    preflight passes on two ranks, the line carries roofline.comm with two ranks' traffic; with a sabotaged collective
    (SIMSEG_BENCH_SABOTAGE_PREFLIGHT=1: rank 1 perturbs what its all-reduce returns) the run ends before anything is timed with a JSON error
    line and a non-zero exit code."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", SIMSEG_DIST_BACKEND="gloo", SIMSEG_BENCH_DEVICE="0", SIMSEG_BENCH_OTHER_DTYPE_LEG="0",
               SIMSEG_BENCH_GELU16_LEG="0")
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--pairs-per-gpu", "32", "--no-seg", "--no-cpu-baseline"]
    bad = subprocess.run(cmd, env=dict(env, SIMSEG_BENCH_SABOTAGE_PREFLIGHT="1"), capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0
    err = json.loads(bad.stdout.strip().splitlines()[-1])
    assert err["value"] is None and "communication preflight failed" in err["error"] and err["n_gpus"] == 2
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["process_group"]["backend"] == "gloo"
    _check_comm_block(line, world=2)


def _worker_seg_sharded(rank, world, port, backend, q):
    import sys
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank if torch.cuda.device_count() >= world else 0), SIMSEG_AMD_COMPUTE="fp32")
    dev = torch.device("cuda", int(os.environ["LOCAL_RANK"]))
    torch.cuda.set_device(dev)
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        from simseg_amd import segpost
        from test_gpu_miou_gate import MEAN, STD, _build, _voc_like
        win, stride, H, W, C, top = 96, 48, 96, 192, 21, 10
        g = torch.Generator().manual_seed(3)
        text = torch.nn.functional.normalize(torch.randn(C, 512, generator=g), dim=-1).to(dev)
        model = _build("vit_test_patch16", 128, "bert-test", 128, win, seed=5).eval().to(dev)
        mean = torch.tensor(MEAN, device=dev).view(1, 3, 1, 1)
        std = torch.tensor(STD, device=dev).view(1, 3, 1, 1)
        batches = []
        for i in range(5):                                   # 5 batches over 2 ranks: 3 + 2 (ragged shard), batch sizes 2 and 1
            b = 2 if i % 2 == 0 else 1
            _, x = _voc_like(b, W, seed=60 + i)
            lab = torch.randint(0, C, (b, H, W), generator=g, dtype=torch.int64).to(torch.uint8)
            batches.append((x[:, :, :H].contiguous(), lab))
        res = segpost.evaluate_sharded(model, batches, text, top, slide=(win, stride), crf=True, mean=mean, std=std, device=dev)
        plain = segpost.evaluate_sharded(model, [(x[:, :, :win, :win].contiguous(), l[:, :win, :win].contiguous()) for x, l in batches], text, top,
                                         slide=None, crf=False, device=dev)
        out = {"hist": res["hist"].cpu(), "images": res["images"], "local": res["images_local"], "miou": float(res["miou"]),
               "plain_hist": plain["hist"].cpu(), "plain_images": plain["images"]}
        if rank == 0:      # the single-process answer: every batch, no group (computed inside the group's process but without using it)
            hist = torch.zeros(3, C, device=dev, dtype=torch.int64)
            hist_p = torch.zeros(3, C, device=dev, dtype=torch.int64)
            with torch.no_grad():
                for x, lab in batches:
                    st = segpost.encode_batch_sliding(model, x.to(dev), text, top, win=win, stride=stride, crf=True, mean=mean, std=std)
                    segpost.finish_batch(st, lab.to(dev), hist=hist)
                    segpost.eval_batch(model, x[:, :, :win, :win].contiguous().to(dev), lab[:, :win, :win].contiguous().to(dev), text, top, hist=hist_p, crf=False)
            out["want"], out["want_plain"] = hist.cpu(), hist_p.cpu()
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_sharded_seg_evaluation_two_ranks_equals_single_process():
    """segpost.evaluate_sharded on two ranks (round-robin batches, ONE all-reduce of the [3, C] area histograms - DESIGN section 5's "seg eval
    shards by image") == the single-process evaluation of every batch: sliding-window form with the DenseCRF and the plain one-input form.
    nccl where two devices are visible, gloo with both ranks on cuda:0 otherwise."""
    world = 2
    backend = "nccl" if torch.cuda.device_count() >= world else "gloo"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_seg_sharded, args=(r, world, port, backend, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(60)
    assert got[0]["images"] == got[1]["images"] == 8 and {got[0]["local"], got[1]["local"]} == {6, 2}
    assert torch.equal(got[0]["hist"], got[1]["hist"]) and torch.equal(got[0]["hist"], got[0]["want"])
    assert int(got[0]["hist"][2].sum()) <= 8 * 96 * 192 and int(got[0]["hist"][1].sum()) > 0
    assert torch.equal(got[0]["plain_hist"], got[0]["want_plain"]) and got[0]["plain_images"] == 8


def _worker_amp_sync(rank, world, port, backend, q):
    """The bench's N > 1 headline step in small: fp16 compute, loss scaled by a live GradScaler, GradSync bucket exchange with the mean left to
    the optimizer kernel, overflow check on the exchanged gradients, fused AdamW - two steps, then one with an inf forced on ONE rank."""
    import sys
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank if torch.cuda.device_count() >= world else 0), SIMSEG_AMD_COMPUTE="fp16", SIMSEG_AMD_TWO_STREAMS="1")
    dev = torch.device("cuda", int(os.environ["LOCAL_RANK"]))
    torch.cuda.set_device(dev)
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        from simseg.utils import ENV
        ENV.rank, ENV.size, ENV.local_rank = rank, world, dev.index
        from simseg_amd.optim import AdamW, GradScaler
        from simseg_amd.parallel import GradSync
        from test_gpu_model import _build
        g = np.load(os.path.join(GOLD, f"clip_train_ws{world}.npz"))

        def golden(name):
            return np.load(os.path.join(GOLD, name + ".npz"))

        batch = {k: torch.from_numpy(g[f"r{rank}.{k}"]).to(dev) for k in ("image", "input_ids", "attention_mask")}
        model = _build(golden).eval()                    # (eval: BERT dropout off - the ranks' functions differ by their data only)
        p0 = {n: p.detach().clone() for n, p in model.named_parameters()}
        sync = GradSync(model.parameters(), overlap=True, average="defer")
        opt = AdamW(model.parameters(), lr=1e-3, half_dtype=torch.float16)
        scaler = GradScaler("cuda", init_scale=1024.0)
        losses = []
        for step in range(3):
            opt.zero_grad(set_to_none=True)
            loss = model(batch)[0]["nce_loss"]
            # step 2: an overflow on ONE rank (its loss blown up before the scaled backward: fp16 gradients saturate) - the exchanged
            # gradient is then non-finite on EVERY rank
            scaler.scale(loss * (1e8 if (step == 2 and rank == 1) else 1.0)).backward()
            sync()
            scaler.step(opt, grad_scale=sync.grad_scale)
            scaler.update()
            losses.append(float(loss.detach()))
        torch.cuda.synchronize()
        taken = opt.steps_taken()
        moved = max(float((p.detach() - p0[n]).abs().max()) for n, p in model.named_parameters())
        flat = torch.cat([p.detach().float().reshape(-1) for p in model.parameters()])
        ref = flat.clone()
        dist.broadcast(ref, src=0)
        q.put((rank, {"losses": losses, "taken": taken, "scale": scaler.get_scale(), "moved": moved, "finite": bool(torch.isfinite(flat).all()),
                      "same_as_rank0": bool(torch.equal(flat, ref))}))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_ranks_fp16_amp_step_with_gradsync():
    """Two ranks run the N > 1 form of the benchmark's headline step (fp16 + GradScaler + GradSync + fused AdamW): both take the same two
    updates and stay bit-identical, the step with an inf injected on rank 1 only is skipped on BOTH ranks (the overflow check reads the
    exchanged gradients) and halves the loss scale on both."""
    world = 2
    backend = "nccl" if torch.cuda.device_count() >= world else "gloo"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_amp_sync, args=(r, world, port, backend, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(60)
    for r in range(world):
        assert got[r]["taken"] == 2 and got[r]["scale"] == 512.0 and got[r]["finite"] and got[r]["moved"] > 0, got[r]
        assert got[r]["same_as_rank0"], "the ranks' parameters diverged"
        assert all(np.isfinite(got[r]["losses"]))


def _worker_rccl_one_rank(port, q):
    """ONE rank on the `nccl` backend (= RCCL; it refuses two ranks per device but serves a one-rank group): the training step with every
    collective of the N > 1 data path issued (SIMSEG_FORCE_COLLECTIVES=1) against the same step with the one-rank shortcuts."""
    import sys
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", SIMSEG_AMD_COMPUTE="fp16",
                      SIMSEG_AMD_TWO_STREAMS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        from simseg.utils import ENV
        ENV.rank, ENV.size, ENV.local_rank = 0, 1, 0
        from simseg_amd.optim import AdamW, GradScaler
        from simseg_amd.parallel import GradSync
        from test_gpu_model import _build
        g = np.load(os.path.join(GOLD, "clip_train_ws2.npz"))

        def golden(name):
            return np.load(os.path.join(GOLD, name + ".npz"))

        batch = {k: torch.from_numpy(g[f"r0.{k}"]).cuda() for k in ("image", "input_ids", "attention_mask")}
        calls = {}
        for name in ("all_reduce", "all_gather_into_tensor", "reduce_scatter_tensor"):
            real = getattr(dist, name)

            def counted(*a, _real=real, _name=name, **kw):
                calls[_name] = calls.get(_name, 0) + 1
                return _real(*a, **kw)
            setattr(dist, name, counted)
        out = {}
        for force in ("0", "1"):
            os.environ["SIMSEG_FORCE_COLLECTIVES"] = force
            calls.clear()
            model = _build(golden).eval()
            assert model.loss.group is not None and dist.get_backend(model.loss.group) == "nccl"
            sync = GradSync(model.parameters(), overlap=True, average="defer")
            opt = AdamW(model.parameters(), lr=1e-3, half_dtype=torch.float16)
            scaler = GradScaler("cuda", init_scale=1024.0)
            losses = []
            for _ in range(3):
                opt.zero_grad(set_to_none=True)
                loss = model(batch)[0]["nce_loss"]
                scaler.scale(loss).backward()
                sync()
                scaler.step(opt, grad_scale=sync.grad_scale)
                scaler.update()
                losses.append(float(loss.detach()))
            torch.cuda.synchronize()
            out[force] = {"losses": losses, "taken": opt.steps_taken(), "calls": dict(calls), "buckets": len(sync.buckets),
                          "params": torch.cat([p.detach().float().reshape(-1) for p in model.parameters()]).cpu()}
            sync.close()
        q.put(out)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_rccl_one_rank_group_runs_every_collective_of_the_step():
    """Real RCCL on the one-GPU box: with SIMSEG_FORCE_COLLECTIVES=1 a one-rank `nccl` group issues the embedding all-gathers (prefetched on
    the towers' streams, handed over with record_stream), the reduce-scatters of their gradients, the bucketed gradient all-reduces on the
    communication stream behind the producers' events and the int64 MIN / MAX check of the bucket cut - the code the N > 1 benchmark runs -
    and the fp16 AMP step gives the losses and parameters of the same step with the one-rank shortcuts."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_rccl_one_rank, args=(_free_port(), q))
    p.start()
    out = q.get(timeout=600)
    p.join(60)
    plain, forced = out["0"], out["1"]
    assert plain["calls"].get("all_gather_into_tensor", 0) == 0 and plain["calls"].get("reduce_scatter_tensor", 0) == 0
    c = forced["calls"]
    assert c.get("all_gather_into_tensor", 0) >= 6 and c.get("reduce_scatter_tensor", 0) >= 6, c          # 2 per step: image and text embeddings
    # every bucket of every step: the first step's single 64-MiB bucket + the int64 MIN / MAX check of the cut at the tower-stream
    # boundaries + the stream-pure buckets of the two later steps
    assert c.get("all_reduce", 0) >= 1 + 2 + 2 * forced["buckets"] and forced["buckets"] > 1, (c, forced["buckets"])
    assert forced["taken"] == plain["taken"] == 3
    np.testing.assert_allclose(forced["losses"], plain["losses"], rtol=2e-3)
    rel = float((forced["params"] - plain["params"]).norm() / plain["params"].norm())
    assert rel < 1e-4, rel
