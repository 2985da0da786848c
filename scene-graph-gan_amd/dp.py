"""Data-parallel training over the GPUs of one node: one process per GPU, gradients all-reduced with RCCL
(torch.distributed backend "nccl" on ROCm) over xGMI, overlapped with the next encoder forward.

The reference is single-device (train.py:417-418); this layer is new.  The path shards naturally: no op couples
samples (LayerNorm is per sample, every loss term is a batch mean, train.py:245-250), so with equal shards the
mean of the per-rank gradients equals the global-batch gradient.  Parameters and Adam slots are replicated.

Collective: all-reduce(sum) of the live range of the flat gradient arena (params.py) in a few large buckets
(xGMI is point-to-point, ring collectives are per-link bound: few, large messages), scaled by 1/world inside the
fused Adam kernel.  Overlap (step.py): the Adam step of a network is deferred until its weights are next needed, and
the network whose reduce is NOT in flight is issued first: the critic-gradient all-reduce runs under G's forward (of
the generator step, or of the next critic update when critic_iters > 1), the generator-gradient all-reduce under the
next critic step's D-encoder forward.  tests/test_dp_order.py pins that order with a recording reducer.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


class PendingReduce:
    def __init__(self, works, scale):
        self.works, self.scale = works, scale

    def wait(self):
        for w in self.works:
            w.wait()           # the current stream waits for the collective (no host block with NCCL/RCCL)
        return self.scale


class GradReducer:
    """Callable used by GanStep: reducer(network) -> PendingReduce."""

    def __init__(self, group=None, bucket_bytes=64 << 20):
        self.group = group
        self.world = dist.get_world_size(group)
        self.bucket_elems = max(1, bucket_bytes // 4)

    def __call__(self, net):
        flat = net.arena.live(net.grad_flat)
        works = []
        for s in range(0, flat.numel(), self.bucket_elems):
            works.append(dist.all_reduce(flat[s:s + self.bucket_elems], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        return PendingReduce(works, 1.0 / self.world)


def init_from_env(backend=None):
    """Initialise torch.distributed from the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # "nccl" is RCCL on ROCm. SGG_DP_BACKEND=gloo lets several ranks share one GPU (rehearsal on a 1-GPU box).
            backend = os.environ.get("SGG_DP_BACKEND", "nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            local = local % torch.cuda.device_count()
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    elif torch.cuda.is_available():
        local = local % torch.cuda.device_count()
    return rank, world, local


def shard_rows(t, rank, world):
    """Rank r takes rows [r*B, (r+1)*B) of a global seeded draw, so N ranks reproduce the single-process batch."""
    B = t.shape[0] // world
    return t[rank * B:(rank + 1) * B]
