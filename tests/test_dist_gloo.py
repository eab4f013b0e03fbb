"""world_size-2 tests of the multi-process path on CPU (gloo): the differentiable embedding gather (all-gather forward,
reduce-scatter-equivalent backward), group construction and the plain gather helpers.  The loss arithmetic between the
collectives is the oracle's here (the HIP loss kernel needs a GPU); what is pinned is the exchange step, against the
gradients the REFERENCE's GatherLayer + NCE produced in a 2-process run (tests/golden/clip_train_ws2.npz)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLD, REPO


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import simseg_ref as R
    from simseg.utils import ENV, GatherLayer, all_gather, concat_all_gather
    from simseg.utils.dist import generate_local_groups
    ENV.rank, ENV.size, ENV.local_rank = rank, world, rank
    g = np.load(os.path.join(GOLD, f"clip_train_ws{world}.npz"))
    f1 = torch.from_numpy(g[f"r{rank}.nce_f1"]).requires_grad_(True)
    f2 = torch.from_numpy(g[f"r{rank}.nce_f2"]).requires_grad_(True)
    ign = torch.from_numpy(g[f"r{rank}.nce_ign"])
    temp = torch.tensor(0.02, requires_grad=True)
    group, grank = generate_local_groups(world)
    assert grank == rank
    f2g = GatherLayer.apply(f2, group, grank)
    ign_g = torch.cat(all_gather(ign))
    loss, acc = R.nce_global(f1, f2g, temp, grank, ign, ign_g)
    loss.backward()
    res = dict(loss=loss.item(), acc=acc.item(), g1=f1.grad.numpy(), g2=f2.grad.numpy(), gt=temp.grad.item(),
               want_loss=float(g[f"r{rank}.nce_loss"]), want_acc=float(g[f"r{rank}.nce_acc"]), want_g1=g[f"r{rank}.nce_g1"],
               want_g2=g[f"r{rank}.nce_g2"], want_gt=float(g[f"r{rank}.nce_gt"]))
    # sub-groups of size 1: every rank is alone, the gather is the identity
    sub, sub_rank = generate_local_groups(1)
    res["sub"] = (dist.get_world_size(sub), sub_rank)
    res["cat"] = concat_all_gather(torch.full((2,), float(rank))).tolist()
    # the reference's other helpers (dist.py:28-40, 124-139, 225-290)
    from simseg.utils.dist import all_gather_with_grad, broadcast_list, broadcast_object_list
    ENV.device = torch.device("cpu")
    t = torch.full((3,), float(rank + 1), requires_grad=True)
    parts = all_gather_with_grad.apply(t)
    sum(float(i + 1) * p.sum() for i, p in enumerate(parts)).backward()
    res["awg"] = ([p[0].item() for p in parts], t.grad.tolist())
    res["blist"] = broadcast_list([rank, 7 + rank], src=1)
    objs = [{"rank": rank}, "x" * (rank + 1)]
    broadcast_object_list(objs, src=0)
    res["bobj"] = objs
    # hook-free gradient synchronisation (simseg_amd/parallel.py): flat all-reduce, grads re-pointed at views of the flat buffer
    from simseg_amd.parallel import GradSync
    torch.manual_seed(0)
    lin = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2))
    lin[1].bias.requires_grad_(False)
    sync = GradSync(lin.parameters())
    x = torch.full((5, 4), float(rank + 1))
    lin(x).square().sum().backward()
    local = [p.grad.clone() for p in lin.parameters() if p.requires_grad]
    sync()
    res["gsync"] = ([g_.tolist() for g_ in local], [p.grad.tolist() for p in lin.parameters() if p.requires_grad],
                    all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(sync.params, sync.views)))
    # the same exchange without overlap (one flat all-reduce after the backward) and with two small buckets: identical means
    # (average="defer": the sum is delivered and the division by the world size is handed to the optimizer kernel as grad_scale)
    for kw in (dict(overlap=False), dict(overlap=True, bucket_mb=0), dict(overlap=True, average="defer")):
        lin2 = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2))
        lin2.load_state_dict(lin.state_dict())
        s2 = GradSync(lin2.parameters(), **kw)
        for _ in range(2):                                   # second step: gradients dropped and re-delivered through the hooks
            for p_ in lin2.parameters():
                p_.grad = None
            lin2(x).square().sum().backward()
            s2.finish()
        assert s2.grad_scale == (0.5 if kw.get("average") == "defer" else 1.0)
        res.setdefault("gsync_alt", []).append([(p.grad * s2.grad_scale).tolist() for p in lin2.parameters()])
    # a second backward before finish(): the overlapped exchange refuses it (its buckets are already reduced) instead of racing
    s3 = GradSync(lin.parameters(), overlap=True)
    lin(x).square().sum().backward()
    try:
        lin(x).square().sum().backward()
        res["gsync_double"] = "no error"
    except RuntimeError as e:
        res["gsync_double"] = str(e)
    for h in s3._handles:
        h.remove()
    # multi-rank retrieval evaluation: uneven shards padded with image_id = -1, gathered, padding dropped (retrieval_evaluation.py:89-95)
    from simseg_amd.retrieval import gather_retrieval_sets
    gen = torch.Generator().manual_seed(5)
    n_all, P = 12, 8
    full = {"image_embeddings": torch.randn(n_all, P, generator=gen), "text_embeddings": torch.randn(n_all, P, generator=gen),
            "image_id": torch.arange(n_all) // 3, "caption_id": torch.arange(n_all)}
    lo, hi = (0, 7) if rank == 0 else (7, 12)                    # 7 and 5 rows
    got = gather_retrieval_sets({k: v[lo:hi] for k, v in full.items()})
    res["retr_gather"] = all(torch.equal(got[k], full[k]) for k in full)
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_gather_layer_matches_reference_two_ranks(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        o = out[r]
        np.testing.assert_allclose(o["loss"], o["want_loss"], rtol=2e-5)
        np.testing.assert_allclose(o["acc"], o["want_acc"], atol=1e-6)
        np.testing.assert_allclose(o["g1"], o["want_g1"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(o["g2"], o["want_g2"], rtol=1e-4, atol=1e-6)     # summed over both ranks' losses
        np.testing.assert_allclose(o["gt"], o["want_gt"], rtol=1e-4)
        assert o["sub"] == (1, 0)
        assert o["cat"] == [0.0, 0.0, 1.0, 1.0]
        assert o["awg"] == ([1.0, 2.0], [float(r + 1)] * 3)          # gradient of this rank's slot only
        assert o["blist"] == [1, 8]
        assert o["bobj"] == [{"rank": 0}, "x"]
        assert o["gsync"][2]
        assert "exactly one backward per finish()" in o["gsync_double"], o["gsync_double"]
        assert o["retr_gather"]
    for r in range(world):                                       # flat / multi-bucket variants agree with the default
        for alt in out[r]["gsync_alt"]:
            ref = [g_ for g_ in out[r]["gsync"][1]]
            got = [a for a, p_ok in zip(alt, [True, True, True, False]) if p_ok]
            for a, b in zip(got, ref):
                np.testing.assert_allclose(np.array(a), np.array(b), rtol=1e-6, atol=1e-7)
    for k in range(len(out[0]["gsync"][0])):                     # every rank ends with the mean of the two ranks' local gradients
        mean = (np.array(out[0]["gsync"][0][k]) + np.array(out[1]["gsync"][0][k])) / 2
        for r in range(world):
            np.testing.assert_allclose(np.array(out[r]["gsync"][1][k]), mean, rtol=1e-6, atol=1e-7)


# ---- four ranks: the whole-world loss and cfg.loss.group_size sub-groups (mml_loss.py:24-27) -------------------------------------------
def _worker_groups(rank, world, port, q, group_size, fixture):
    """The exchange step on `world` ranks with the loss gathered over sub-groups of `group_size` ranks: this repo's generate_local_groups
    (host-by-host packing, new_group on every rank) + GatherLayer (all-gather forward, reduce-scatter backward over the SUB-group) around
    the oracle's loss arithmetic, against what the REFERENCE's own generate_local_groups + GatherLayer + NCE produced in a 4-process gloo
    run (oracle/make_golden.py dist4 / dist4g2)."""
    import sys
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import simseg_ref as R
    from simseg.utils import ENV, GatherLayer
    from simseg.utils.dist import all_gather_group, generate_local_groups
    ENV.rank, ENV.size, ENV.local_rank = rank, world, rank
    g = np.load(os.path.join(GOLD, fixture + ".npz"))
    group, grank = generate_local_groups(group_size)
    res = {"group": (dist.get_world_size(group), grank, dist.get_process_group_ranks(group))}
    f1 = torch.from_numpy(g[f"r{rank}.nce_f1"]).requires_grad_(True)
    f2 = torch.from_numpy(g[f"r{rank}.nce_f2"]).requires_grad_(True)
    ign = torch.from_numpy(g[f"r{rank}.nce_ign"])
    temp = torch.tensor(0.02, requires_grad=True)
    f2g = GatherLayer.apply(f2, group, grank)
    ign_g = torch.cat(all_gather_group(ign, group))
    loss, acc = R.nce_global(f1, f2g, temp, grank, ign, ign_g)
    loss.backward()
    res.update(loss=loss.item(), acc=acc.item(), g1=f1.grad.numpy(), g2=f2.grad.numpy(), gt=temp.grad.item(), rows=f2g.shape[0],
               want_loss=float(g[f"r{rank}.nce_loss"]), want_acc=float(g[f"r{rank}.nce_acc"]), want_g1=g[f"r{rank}.nce_g1"],
               want_g2=g[f"r{rank}.nce_g2"], want_gt=float(g[f"r{rank}.nce_gt"]))
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def _spawn(target, world, *args):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + args) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return out


@pytest.mark.parametrize("group_size,fixture", [(4, "clip_train_ws4"), (2, "clip_train_ws4g2")])
def test_gather_layer_and_local_groups_match_reference_four_ranks(group_size, fixture):
    world = 4
    out = _spawn(_worker_groups, world, group_size, fixture)
    for r in range(world):
        o = out[r]
        first = r // group_size * group_size
        assert o["group"] == (group_size, r % group_size, list(range(first, first + group_size))), o["group"]
        assert o["rows"] == 8 * group_size                       # the loss sees its sub-group's rows only
        np.testing.assert_allclose(o["loss"], o["want_loss"], rtol=2e-5)
        np.testing.assert_allclose(o["acc"], o["want_acc"], atol=1e-6)
        np.testing.assert_allclose(o["g1"], o["want_g1"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(o["g2"], o["want_g2"], rtol=1e-4, atol=1e-6)     # summed over the sub-group's losses
        np.testing.assert_allclose(o["gt"], o["want_gt"], rtol=1e-4)


# ---- the same global batch on 1 / 2 / 4 ranks gives the same loss and the same averaged gradients -------------------
def _invariance_case():
    gen = torch.Generator().manual_seed(17)
    x1, x2 = torch.randn(8, 24, generator=gen), torch.randn(8, 24, generator=gen)
    w1, w2 = torch.randn(16, 24, generator=gen) * 0.3, torch.randn(16, 24, generator=gen) * 0.3
    return x1, x2, w1, w2


def _invariance_step(x1, x2, w1, w2, temp, gather, rank):
    from oracle import simseg_ref as R
    e1 = torch.nn.functional.normalize(x1 @ w1.T, dim=-1)
    e2 = torch.nn.functional.normalize(x2 @ w2.T, dim=-1)
    l1, _ = R.nce_global(e1, gather(e2), temp, rank)
    l2, _ = R.nce_global(e2, gather(e1), temp, rank)
    return 0.5 * (l1 + l2)


def _worker_invariance(rank, world, port, q):
    import sys
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from simseg.utils import ENV, GatherLayer
    from simseg.utils.dist import generate_local_groups
    ENV.rank, ENV.size, ENV.local_rank = rank, world, rank
    group, grank = generate_local_groups(world)
    x1, x2, w1, w2 = _invariance_case()
    n = 8 // world
    w1, w2 = w1.clone().requires_grad_(True), w2.clone().requires_grad_(True)
    temp = torch.tensor(0.05, requires_grad=True)
    loss = _invariance_step(x1[rank * n:(rank + 1) * n], x2[rank * n:(rank + 1) * n], w1, w2, temp, lambda t: GatherLayer.apply(t, group, grank), grank)
    loss.backward()
    vals = [loss.detach().clone(), w1.grad, w2.grad, temp.grad]
    for v in vals:                                               # what DDP / GradSync do with parameter gradients (core/hooks/dist.py:48-51)
        dist.all_reduce(v)
        v /= world
    q.put((rank, [v.numpy() for v in vals]))
    dist.barrier()
    dist.destroy_process_group()


def test_loss_and_gradients_do_not_depend_on_the_number_of_ranks():
    x1, x2, w1, w2 = _invariance_case()
    w1, w2 = w1.clone().requires_grad_(True), w2.clone().requires_grad_(True)
    temp = torch.tensor(0.05, requires_grad=True)
    loss = _invariance_step(x1, x2, w1, w2, temp, lambda t: t, 0)
    loss.backward()
    want = [loss.detach().numpy(), w1.grad.numpy(), w2.grad.numpy(), temp.grad.numpy()]
    for world in (2, 4):
        out = _spawn(_worker_invariance, world)
        for r in range(world):
            for got, ref in zip(out[r], want):
                np.testing.assert_allclose(got, ref, rtol=2e-5, atol=1e-6, err_msg=f"world {world} rank {r}")


# ---- round 6: the communication preflight and the bench line's roofline.comm block (simseg_amd/commcheck.py) ------------------------------
def _commcheck_worker(rank, world, port, q, sabotage):
    import sys
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from simseg_amd import commcheck, heads
    from simseg_amd.parallel import GradSync
    dev = torch.device("cpu")
    res = {}
    if sabotage and rank == 1:          # one rank's all-reduce returns garbage: EVERY rank's preflight must raise
        real = dist.all_reduce

        def broken(t, *a, **kw):
            w = real(t, *a, **kw)
            if t.dtype == torch.float32:
                t.add_(1.0)
            return w
        dist.all_reduce = broken
    try:
        probe = commcheck.preflight(dev, 8, bucket_bytes=1 << 16, iters=2,
                                    ranks_info=[{"rank": r, "device": f"cpu:{r}"} for r in range(world)])
        res["probe"] = probe
    except commcheck.PreflightError as e:
        res["error"] = str(e)
    if not sabotage:
        # the local stand-ins of the bench's communication-free leg: same shapes, nothing exchanged
        x = torch.full((3, 4), float(rank + 1), requires_grad=True)
        heads.LOCAL_STANDIN = True
        try:
            gathered = heads.GatherLayer.apply(x, dist.group.WORLD, rank)
            (gathered * torch.arange(1, world * 3 + 1, dtype=torch.float32)[:, None]).sum().backward()
        finally:
            heads.LOCAL_STANDIN = False
        res["standin"] = (gathered.detach().clone(), x.grad.clone())
        lin = torch.nn.Linear(4, 3)
        sync = GradSync(lin.parameters())
        sync.skip_collectives = True
        lin(torch.full((2, 4), float(rank + 1))).sum().backward()
        sync()
        res["skip_grad"] = lin.weight.grad.clone()
        sync.close()
        res["block"] = commcheck.comm_roofline(world, 8, 1000 * 4, 3, probe, 90.0, 86.5, process_group={"backend": "gloo"})
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("sabotage", [False, True])
def test_comm_preflight_and_roofline_block_two_ranks(sabotage):
    """Every collective of the step once with a value check (all-gather / reduce-scatter of the embeddings, a gradient bucket's all-reduce,
    the int64 MIN / MAX check), the same verdict on every rank - also when only ONE rank's collective returns wrong values; the local
    stand-ins of the communication-free leg keep the shapes and exchange nothing; the roofline.comm arithmetic."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_commcheck_worker, args=(r, world, port, q, sabotage)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if sabotage:
        assert all("error" in got[r] for r in range(world)), got
        assert "wrong values" in got[1]["error"] and "another rank" in got[0]["error"]
        return
    for r in range(world):
        pr = got[r]["probe"]
        assert set(pr) == {"C1_embedding_all_gather", "C2_embedding_grad_reduce_scatter", "C4_gradient_bucket_all_reduce", "bucket_cut_check_int64_min_max"}
        assert all(v["ok"] for v in pr.values())
        assert pr["C1_embedding_all_gather"]["wire_bytes_per_rank"] == 8 * 512 * 4                 # (W - 1) blocks
        assert pr["C2_embedding_grad_reduce_scatter"]["wire_bytes_per_rank"] == 8 * 512 * 4         # (W - 1) / W of W blocks
        assert pr["C4_gradient_bucket_all_reduce"]["wire_bytes_per_rank"] == (1 << 16)              # 2 (W - 1) / W
        g, gx = got[r]["standin"]
        assert g.shape == (6, 4) and torch.all(g == float(r + 1))                                    # this rank's rows, twice
        want = torch.arange(1, 7, dtype=torch.float32).view(2, 3)[r]
        assert torch.equal(gx, want[:, None].expand(3, 4))                                           # this rank's slice of the gradient, no sum
        assert torch.all(got[r]["skip_grad"] == 2.0 * (r + 1) / world)                               # the LOCAL gradient (not the sum over ranks) times 1 / W
        blk = got[r]["block"]
        assert blk["exposed_communication_ms_per_step"] == 3.5 and blk["world_size"] == 2 and blk["peak"] == 7 * 153.0
        assert blk["per_class"]["C4_gradient_all_reduce"]["wire_bytes_per_rank_step"] == 4000 and blk["per_class"]["C4_gradient_all_reduce"]["per_step"] == 3
        assert blk["wire_bytes_per_rank_step"] == 4000 + 2 * 8 * 512 * 4 + 2 * 8 * 512 * 4
        assert blk["achieved"] == round(blk["wire_bytes_per_rank_step"] / 3.5e-3 / 1e9, 1)
