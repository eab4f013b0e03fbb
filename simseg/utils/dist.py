"""torch.distributed helpers with the reference's names (simseg/utils/dist.py).  backend 'nccl' is RCCL over xGMI on
ROCm; every helper degrades to the identity when the process group is not initialised (single process)."""
import os

import torch
import torch.distributed as dist

from simseg_amd.heads import GatherLayer  # differentiable all-gather (reduce-scatter backward)

from . import logger
from .context import ENV

__all__ = ["all_gather", "all_gather_group", "all_reduce", "broadcast", "barrier", "GatherLayer", "concat_all_gather",
           "generate_local_groups", "all_gather_object", "all_gather_with_grad", "broadcast_list", "broadcast_object_list"]


def _on():
    return dist.is_available() and dist.is_initialized()


def all_gather(tensor, group=None):
    """List of every rank's tensor (equal shapes), dist.py:43-62."""
    if not _on() or dist.get_world_size(group) == 1:
        return [tensor]
    out = [torch.empty_like(tensor) for _ in range(dist.get_world_size(group))]
    dist.all_gather(out, tensor.contiguous(), group=group)
    return out


def all_gather_group(tensor, group):
    """dist.py:65-74: non-differentiable gather inside a sub-group."""
    return all_gather(tensor, group)


def all_reduce(tensor, op="sum", group=None):
    """dist.py:77-102; op in {'sum', 'mean', 'max', 'min'}; in place, returns the tensor."""
    if not _on():
        return tensor
    ops = {"sum": dist.ReduceOp.SUM, "mean": dist.ReduceOp.SUM, "max": dist.ReduceOp.MAX, "min": dist.ReduceOp.MIN}
    dist.all_reduce(tensor, op=ops[op], group=group)
    if op == "mean":
        tensor /= dist.get_world_size(group)
    return tensor


def broadcast(tensor, src=0, group=None):
    if _on():
        dist.broadcast(tensor, src, group=group)
    return tensor


def barrier(group=None):
    if _on():
        dist.barrier(group=group)


def all_gather_object(object_list, obj, group=None, *_unused):
    """dist.py:165-222 pickles through a device tensor; torch's own object collective does the same job."""
    if not _on():
        object_list[0] = obj
        return
    dist.all_gather_object(object_list, obj, group=group)


class all_gather_with_grad(torch.autograd.Function):
    """dist.py:28-40: tuple of every rank's tensor; the backward keeps the gradient of this rank's slot only (no reduction:
    the older of the reference's two differentiable gathers - NCE uses GatherLayer)."""

    @staticmethod
    def forward(ctx, tensor):
        if not _on():
            return (tensor.clone(),)
        out = [torch.empty_like(tensor) for _ in range(dist.get_world_size())]
        dist.all_gather(out, tensor.contiguous())
        return tuple(out)

    @staticmethod
    def backward(ctx, *grads):
        return grads[dist.get_rank() if _on() else 0].clone()


def broadcast_list(x, name=None, src=0):
    """dist.py:124-139: a list of numbers through a device tensor; returns the source rank's list (not in place)."""
    if not isinstance(x, list):
        raise AssertionError("broadcast_list only takes list as input.")
    t = torch.tensor(x, device=ENV.device if ENV.device is not None else "cpu")
    broadcast(t, src=src)
    return t.tolist()


def broadcast_object_list(object_list, src=0, group=None):
    """dist.py:225-290 re-implements torch's object broadcast; torch's own does the job (objects pickled through tensors on the
    current device for RCCL)."""
    if _on():
        dist.broadcast_object_list(object_list, src=src, group=group)


@torch.no_grad()
def concat_all_gather(tensor):
    return torch.cat(all_gather(tensor), dim=0)


def generate_local_groups(local_group_size):
    """dist.py:371-427: partition the world into groups of `local_group_size` ranks, filled host by host, and return
    (this rank's group, this rank's index inside it).  Every rank creates every group (new_group is collective)."""
    if not _on():
        logger.error("this function is only supported by pytorch distributed training")
        raise SystemExit(1)
    world, me = dist.get_world_size(), dist.get_rank()
    if world % local_group_size != 0:
        raise AssertionError(f"world size {world} is not a multiple of loss.group_size {local_group_size}")
    if local_group_size == world:
        return dist.group.WORLD, me
    infos = [None] * world
    dist.all_gather_object(infos, (os.environ.get("HOSTNAME", "localhost"), me))
    by_host = {}
    for host, r in infos:
        by_host.setdefault(host, []).append(r)
    groups, leftovers = [], []
    for host in by_host:                       # whole groups inside one host first
        ranks = by_host[host]
        while len(ranks) >= local_group_size:
            groups.append(ranks[:local_group_size])
            ranks = ranks[local_group_size:]
        leftovers.extend(ranks)
    while leftovers:                           # then pack what is left across hosts
        groups.append(leftovers[:local_group_size])
        leftovers = leftovers[local_group_size:]
    mine, my_idx = None, 0
    for ranks in groups:
        g = dist.new_group(ranks)
        if me in ranks:
            mine, my_idx = g, ranks.index(me)
            logger.info(f"Generate a local group {ranks}, local_group_rank {my_idx} for rank {me}", root_only=False)
    return mine, my_idx
