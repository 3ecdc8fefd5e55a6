"""world_size-2 gloo test of the data-parallel path (CPU): the flat-bucket all-reduce of awm_amd.distributed
reproduces the big-batch gradient when the per-rank shards are equal-sized.  The per-rank gradient here comes from
the CPU oracle (the HIP modules need a GPU); what is under test is the sharding + collective + averaging."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _grad_flat(gsd, dsd, s, msg):
    from oracle import wm_oracle as O
    g2 = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in gsd.items()}
    d2 = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in dsd.items()}
    total, _ = O.step_losses(g2, d2, s, msg, training=False)     # eval-mode BN: no cross-clip statistics
    total.backward()
    keys = [(g2, k) for k in g2 if g2[k].requires_grad and k != "embedding.weight"] + [(d2, k) for k in d2 if d2[k].requires_grad]
    return torch.cat([sd[k].grad.reshape(-1) for sd, k in keys])


def _worker(rank, world, port, T, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")      # no hostname lookup (the container's name may not resolve: minutes of DNS timeouts)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    import awm_amd
    from awm_amd import distributed as wmd
    from oracle import recipes as R, wm_oracle as O
    gsd, dsd = R.reference_layout_init()
    R.perturb_bn_(gsd, 7); R.perturb_bn_(dsd, 8)
    # parameter broadcast: rank 1 starts from garbage and must end up with rank 0's replica
    lin = torch.nn.Linear(4, 4)
    if rank == 1:
        with torch.no_grad():
            for p in lin.parameters():
                p.add_(1.0)
    wmd.broadcast_parameters([lin])
    ref = [torch.zeros_like(p) for p in lin.parameters()]
    for r_, p in zip(ref, lin.parameters()):
        r_.copy_(p.data); dist.broadcast(r_, 0)
        assert torch.equal(r_, p.data)
    n_total = 4
    lo, hi = wmd.shard_range(n_total, rank, world)
    s = O.synthetic_clips(n_total, seed=5, T=T)[lo:hi]
    msg = O.synthetic_messages(n_total, seed=6)[lo:hi]
    flat = _grad_flat(gsd, dsd, s, msg)
    wmd.allreduce_flat_gradient(flat)
    if rank == 0:
        torch.save(flat, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_allreduce_matches_big_batch(tmp_path):
    from oracle import recipes as R, wm_oracle as O
    T = 2000
    out = str(tmp_path / "flat.pt")
    mp.spawn(_worker, args=(2, _free_port(), T, out), nprocs=2, join=True)
    flat2 = torch.load(out)
    gsd, dsd = R.reference_layout_init()
    R.perturb_bn_(gsd, 7); R.perturb_bn_(dsd, 8)
    big = _grad_flat(gsd, dsd, O.synthetic_clips(4, seed=5, T=T), O.synthetic_messages(4, seed=6))
    err = float((flat2 - big).abs().max() / big.abs().max())
    assert err < 5e-4, f"DP-averaged gradient differs from the big-batch gradient: {err:.2e}"


class _FlatStore:
    """CPU double of optim.FlatAdam's layout (params / grads as views of one flat buffer) for the GradSync test"""

    def __init__(self, modules):
        self.params = [p for m in modules for p in m.parameters()]
        n = sum(p.numel() for p in self.params)
        self.flat, self.grad, self._spans, off = torch.empty(n), torch.zeros(n), [], 0
        for p in self.params:
            k = p.numel()
            self.flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + k].view_as(p)
            p.grad = self.grad[off:off + k].view_as(p)
            self._spans.append((off, k)); off += k


def _gradsync_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")      # no hostname lookup (the container's name may not resolve: minutes of DNS timeouts)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from awm_amd import distributed as wmd
    torch.manual_seed(3)
    gen, det = torch.nn.Linear(6, 5), torch.nn.Sequential(torch.nn.Linear(5, 4), torch.nn.Linear(4, 1))
    store = _FlatStore([gen, det])
    sync = wmd.GradSync(store, early_modules=[det])
    assert sync.early == (store._spans[2][0], 5 * 4 + 4 + 4 + 1)
    x = torch.randn(8, 6, generator=torch.Generator().manual_seed(11))
    lo, hi = wmd.shard_range(8, rank, world)
    for _ in range(2):                                   # two rounds: the hook countdown must re-arm
        store.grad.zero_()
        det(gen(x[lo:hi])).mean().backward()
        assert sync._work is not None, "the early (Detector) span was not launched from the accumulate hooks"
        sync()
    if rank == 0:
        torch.save(store.grad.clone(), out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_gradsync_early_bucket_two_ranks(tmp_path):
    """GradSync: Detector span all-reduced from the post-accumulate hooks while backward is still running, the rest after
    backward; the result equals the big-batch gradient (equal shards, mean loss)"""
    out = str(tmp_path / "g.pt")
    mp.spawn(_gradsync_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    torch.manual_seed(3)
    gen, det = torch.nn.Linear(6, 5), torch.nn.Sequential(torch.nn.Linear(5, 4), torch.nn.Linear(4, 1))
    x = torch.randn(8, 6, generator=torch.Generator().manual_seed(11))
    det(gen(x)).mean().backward()
    want = torch.cat([p.grad.reshape(-1) for m in (gen, det) for p in m.parameters()])
    assert float((got - want).abs().max()) < 1e-6


def test_shard_range_covers_everything():
    from awm_amd import distributed as wmd
    for n in (1, 7, 256, 1000):
        for w in (1, 2, 3, 8):
            spans = [wmd.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
