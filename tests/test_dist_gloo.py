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
    for kw in (dict(overlap=False), dict(overlap=True, bucket_mb=0)):
        lin2 = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2))
        lin2.load_state_dict(lin.state_dict())
        s2 = GradSync(lin2.parameters(), **kw)
        for _ in range(2):                                   # second step: gradients dropped and re-delivered through the hooks
            for p_ in lin2.parameters():
                p_.grad = None
            lin2(x).square().sum().backward()
            s2.finish()
        res.setdefault("gsync_alt", []).append([p.grad.tolist() for p in lin2.parameters()])
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
