"""Data parallelism over the clip batch: one process per GPU, parameters replicated, ONE sum-all-reduce of
the flat fp32 gradient bucket per step (RCCL over xGMI; backend "nccl" is RCCL on ROCm).  The reference is
single-GPU: clips are independent units, BatchNorm statistics and the loss means are per
rank (DDP semantics), so averaging the gradients reproduces the big-batch mean for equal-sized shards.
With gloo the same code runs on CPU tensors for the world_size-2 tests."""
from __future__ import annotations

import torch
import torch.distributed as dist


def broadcast_parameters(modules, src=0):
    """initial parameter + buffer broadcast from rank `src` (identical replicas on every GPU)"""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    for m in modules:
        for t in list(m.parameters()) + list(m.buffers()):
            dist.broadcast(t.data, src)


def allreduce_flat_gradient(flat_grad: torch.Tensor):
    """sum over ranks then divide by the world size: 17.5 MB for main16, one bucket, one collective"""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return flat_grad
    dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
    flat_grad.div_(dist.get_world_size())
    return flat_grad


def allreduce_gradients(params):
    """same for parameters that do not live in a flat buffer: pack -> one all-reduce -> unpack"""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    grads = [p.grad for p in params if p.grad is not None]
    flat = torch.cat([g.reshape(-1) for g in grads])
    allreduce_flat_gradient(flat)
    off = 0
    for g in grads:
        g.copy_(flat[off:off + g.numel()].view_as(g))
        off += g.numel()


def shard_range(n_units: int, rank: int, world: int):
    """contiguous, near-equal split of independent units (clips) over ranks"""
    base, rem = divmod(n_units, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
