"""Data parallelism over the clip batch: one process per GPU, parameters replicated, sum-all-reduce of the flat fp32
gradient bucket each step (RCCL over xGMI; backend "nccl" is RCCL on ROCm).  The reference is single-GPU
(SURVEY.md 8(e)): clips are independent units, BatchNorm statistics and the loss means are per rank (DDP
semantics), so averaging the gradients reproduces the big-batch mean for equal-sized shards.

The path has no data-path collective; the only exchange is the gradient bucket (17.5 MB for main16, 99.6 MB for
main14b_2 hd=256).  `GradSync` splits it in two spans of the flat buffer: the Detector's gradients are complete long
before the Generator's (backward visits the Detector first, then 8+ ms of LSTM BPTT), so their all-reduce is issued
from a post-accumulate hook on a side stream while backward is still running; the Generator span follows when
backward returns, and Adam waits for both.  With gloo (CPU rehearsal / world-size-2 tests, or several ranks sharing
one GPU) the same code stages the spans through host memory synchronously."""
from __future__ import annotations

import torch
import torch.distributed as dist


def _active():
    return dist.is_initialized() and dist.get_world_size() > 1


def _needs_host_staging(t: torch.Tensor) -> bool:
    return t.is_cuda and dist.get_backend() == "gloo"


def _all_reduce_sum(t: torch.Tensor):
    """in-place sum over ranks; gloo + GPU tensor goes through host memory (rehearsal path)"""
    if _needs_host_staging(t):
        h = t.detach().cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)


def broadcast_parameters(modules, src=0):
    """initial parameter + buffer broadcast from rank `src` (identical replicas on every GPU)"""
    if not _active():
        return
    for m in modules:
        for t in list(m.parameters()) + list(m.buffers()):
            if _needs_host_staging(t.data):
                h = t.data.cpu()
                dist.broadcast(h, src)
                t.data.copy_(h)
            else:
                dist.broadcast(t.data, src)


def allreduce_flat_gradient(flat_grad: torch.Tensor):
    """sum over ranks then divide by the world size: 17.5 MB for main16, one bucket, one collective"""
    if not _active():
        return flat_grad
    _all_reduce_sum(flat_grad)
    flat_grad.div_(dist.get_world_size())
    return flat_grad


def allreduce_gradients(params):
    """same for parameters that do not live in a flat buffer: pack -> one all-reduce -> unpack"""
    if not _active():
        return
    grads = [p.grad for p in params if p.grad is not None]
    flat = torch.cat([g.reshape(-1) for g in grads])
    allreduce_flat_gradient(flat)
    off = 0
    for g in grads:
        g.copy_(flat[off:off + g.numel()].view_as(g))
        off += g.numel()


class GradSync:
    """Gradient exchange for optim.FlatAdam: `early_modules` (the Detector) own a contiguous span of the flat gradient
    whose all-reduce is launched as soon as autograd has accumulated the last of their gradients; __call__ (the
    `grad_sync` argument of train_step, after backward) reduces the rest, joins and divides by the world size.
    `force=True` runs the same code at world size 1 (a one-rank RCCL communicator), for rehearsing on a one-GPU box."""

    def __init__(self, opt, early_modules=(), force=False):
        self.opt = opt
        self.force = bool(force)
        self.early = None                 # (offset, length) of the early span inside opt.grad
        self._pending = 0
        self._n_early = 0
        self._work = None
        self._side = None
        self._handles = []
        ids = {id(p): i for i, p in enumerate(opt.params)}
        early = [p for m in early_modules for p in m.parameters()]
        if early and not any(getattr(p, "_wm_grad", None) is not None for p in opt.params):
            idx = sorted(ids[id(p)] for p in early)
            if idx == list(range(idx[0], idx[-1] + 1)):          # contiguous in the flat buffer
                off = opt._spans[idx[0]][0]
                self.early = (off, sum(opt._spans[i][1] for i in idx))
                self._n_early = len(early)
                for p in early:
                    self._handles.append(p.register_post_accumulate_grad_hook(self._on_accumulated))
        self._pending = self._n_early

    def enabled(self):
        return self.force or _active()

    def begin_step(self):
        """re-arm for a new backward (train_step calls it with zero_grad): a backward whose __call__ never ran -- an exception
        between the two, or a second backward -- must not leave its collective or its hook count behind"""
        if self._work is not None:
            self._work.wait()
            if self._side is not None:
                torch.cuda.current_stream().wait_stream(self._side)
            self._work = None
        self._pending = self._n_early

    def _on_accumulated(self, _param):
        if not self.enabled():                        # single process: nothing to exchange, nothing to count
            return
        self._pending -= 1
        if self._pending < 0 or (self._pending == 0 and self._work is not None):
            raise RuntimeError("GradSync: a second backward reached the early-span hooks before the first one was exchanged "
                               "(call the GradSync object after every backward, or begin_step() to discard one)")
        if self._pending == 0 and self.enabled():
            self._launch_early()

    def _launch_early(self):
        off, n = self.early
        span = self.opt.grad[off:off + n]
        if _needs_host_staging(span):
            return                                    # gloo + GPU tensors (rehearsal): reduced synchronously in __call__
        if not span.is_cuda:                          # CPU tensors (gloo tests): plain async collective
            self._work = dist.all_reduce(span, op=dist.ReduceOp.SUM, async_op=True)
            return
        if self._side is None:
            self._side = torch.cuda.Stream()
        main = torch.cuda.current_stream()
        self._side.wait_stream(main)                  # every accumulate kernel enqueued so far precedes the collective
        with torch.cuda.stream(self._side):
            self._work = dist.all_reduce(span, op=dist.ReduceOp.SUM, async_op=True)

    def __call__(self):
        if not self.enabled():
            self._pending = self._n_early
            return
        g = self.opt.grad
        if self.early is not None and self._work is not None:
            off, n = self.early
            # the rest of the bucket: [0, off) and [off + n, end) -- the Generator's parameters
            for lo, hi in ((0, off), (off + n, g.numel())):
                if hi > lo:
                    _all_reduce_sum(g[lo:hi])
            self._work.wait()                         # orders the side-stream collective before the current stream
            if self._side is not None:
                torch.cuda.current_stream().wait_stream(self._side)
            self._work = None
        else:
            _all_reduce_sum(g)
        g.div_(dist.get_world_size())
        self._pending = self._n_early

    def close(self):
        for h in self._handles:
            h.remove()
        self._handles = []


def shard_range(n_units: int, rank: int, world: int):
    """contiguous, near-equal split of independent units (clips) over ranks"""
    base, rem = divmod(n_units, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
