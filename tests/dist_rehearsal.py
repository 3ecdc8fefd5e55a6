#!/usr/bin/env python
"""Rank program of the multi-GPU rehearsal (tests/test_gpu_distributed.py): the REAL HIP modules + optim.FlatAdam +
distributed.GradSync under world_size ranks that share one GPU, gloo as the transport (RCCL refuses two ranks on one
device).  Started as a fresh child process per rank -- never from a process that has touched the GPU.

  phase 1 (eval-mode BatchNorm, no cross-clip statistics): shard gradient -> GradSync -> rank 0 saves the averaged
          flat gradient (must equal the single-process gradient of the concatenated batch);
  phase 2 (train mode): two train_step()s with the gradient exchange -> every rank saves its flat parameter buffer
          (replicas must stay identical)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(model, dev):
    """(G, D, forward_losses, train_step) of the requested model family, identical on every rank before the broadcast"""
    import awm_amd
    if model == "main14b_2":
        from awm_amd import main14b_2 as M14
        torch.manual_seed(42)
        G, D = M14.Generator(hidden_dim=256), M14.Detector()         # BASELINE configs[4]: 99.6 MB gradient bucket
        return G.to(dev), D.to(dev), M14.forward_losses, M14.train_step
    from oracle import recipes as R
    gsd, dsd = R.reference_layout_init()
    R.perturb_bn_(gsd, R.BN_SEED_G); R.perturb_bn_(dsd, R.BN_SEED_D)
    G, D = awm_amd.Generator(16), awm_amd.Detector(16)
    G.load_state_dict(gsd); D.load_state_dict(dsd)
    return G.to(dev), D.to(dev), awm_amd.forward_losses, awm_amd.train_step


def force_sync_main(out_dir, n_total, T, model):
    """ONE rank on a real RCCL communicator (backend nccl, world size 1): GradSync(force=True) takes the production path --
    post-accumulate hooks count the Detector's gradients down, the early span's all-reduce is launched from the autograd thread
    on the side stream, __call__ reduces the rest, joins and divides by 1.  The exchanged gradient must equal the plain one."""
    import awm_amd
    from awm_amd import distributed as wmd
    from oracle import wm_oracle as O
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    G, D, fwd, train_step = build(model, dev)
    G.train(); D.train()
    opt = awm_amd.FlatAdam([G, D], lr=1e-3)
    s = O.synthetic_clips(n_total, seed=41, T=T).to(dev)
    msg = O.synthetic_messages(n_total, seed=42).to(dev)
    snap = {id(m): {k: v.clone() for k, v in m.state_dict().items()} for m in (G, D)}    # per module: G and D of main14b_2 share key names
    plains = []
    for _ in range(2):                                       # twice: the very first backward of a process must equal the second
        opt.zero_grad()
        fwd(G, D, s, msg)[0].backward()
        opt.finish_backward()
        plains.append(opt.grad.clone())
        for m in (G, D):                                     # BatchNorm running statistics back to where the plain run started
            m.load_state_dict(snap[id(m)])
    plain = plains[1]
    first_equal = torch.equal(plains[0], plains[1])
    first_diff = float((plains[0] - plains[1]).abs().max())
    sync = wmd.GradSync(opt, early_modules=[D], force=True)
    assert sync.early is not None and sync.enabled()
    launched = []
    orig = sync._launch_early

    def spy():
        orig()
        launched.append((sync._work is not None, sync._pending))
    sync._launch_early = spy
    opt.zero_grad(); sync.begin_step()
    fwd(G, D, s, msg)[0].backward()
    assert launched == [(True, 0)], launched                 # the hook launched the async collective during backward
    opt.finish_backward()
    sync()
    torch.cuda.synchronize()
    assert sync._work is None and sync._pending == sync._n_early
    same = torch.equal(opt.grad, plain)
    where = []
    if not same:                                             # name the parameters that differ (diagnostic for the assertion message)
        names = [f"{n}.{k}" for n, m in (("G", G), ("D", D)) for k, _ in m.named_parameters()]
        for (off, k), nm in zip(opt._spans, names):
            d = float((opt.grad[off:off + k] - plain[off:off + k]).abs().max())
            if d > 0:
                where.append((nm, off, k, d, float(plain[off:off + k].abs().max())))
    # a backward that is never exchanged must not poison the next step: begin_step() (train_step calls it) discards it
    opt.zero_grad(); sync.begin_step()
    fwd(G, D, s, msg)[0].backward()
    assert sync._work is not None
    losses = [float(train_step(G, D, opt, s, msg, grad_sync=sync)["total"]) for _ in range(2)]
    assert len(launched) == 4 and sync._work is None
    # and without begin_step a second backward is refused loudly instead of leaving stale state behind
    opt.zero_grad(); sync.begin_step()
    fwd(G, D, s, msg)[0].backward()
    refused = False
    try:
        fwd(G, D, s, msg)[0].backward()
    except RuntimeError as e:
        refused = "second backward" in str(e)
    sync.begin_step()
    torch.cuda.synchronize()
    torch.save({"same": same, "max_diff": float((opt.grad - plain).abs().max()) if not same else 0.0, "losses": losses,
                "refused": refused, "early_span": sync.early, "where": where,
                "first_backward_equals_second": first_equal, "first_vs_second_max_diff": first_diff, "bucket_bytes": 4 * opt.grad.numel()}, os.path.join(out_dir, "force_sync.pt"))
    dist.destroy_process_group()


def main():
    out_dir, n_total, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    model = sys.argv[4] if len(sys.argv) > 4 else "main16"
    if len(sys.argv) > 5 and sys.argv[5] == "force_sync":
        return force_sync_main(out_dir, n_total, T, model)
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")      # no hostname lookup (it may not resolve on the box)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    import awm_amd
    from awm_amd import distributed as wmd
    from oracle import wm_oracle as O
    awm_amd.lib.load()
    G, D, fwd, train_step = build(model, dev)
    if rank != 0:                                    # replicas must come from the broadcast, not from the seed
        with torch.no_grad():
            for p in list(G.parameters()) + list(D.parameters()):
                p.add_(0.5)
    wmd.broadcast_parameters([G, D])
    opt = awm_amd.FlatAdam([G, D], lr=1e-3)
    sync = wmd.GradSync(opt, early_modules=[D])
    assert sync.early is not None
    lo, hi = wmd.shard_range(n_total, rank, world)
    s = O.synthetic_clips(n_total, seed=41, T=T)[lo:hi].to(dev)
    msg = O.synthetic_messages(n_total, seed=42)[lo:hi].to(dev)
    # phase 1 (main16: eval-mode BatchNorm so that no statistic crosses clips; main14b_2 has no BatchNorm)
    if model == "main16":
        G.eval(); D.eval()
    opt.zero_grad()
    total, _ = fwd(G, D, s, msg)
    total.backward()
    opt.finish_backward()
    sync()
    torch.cuda.synchronize()
    if rank == 0:
        torch.save(opt.grad.cpu(), os.path.join(out_dir, "avg_grad.pt"))
    # phase 2
    G.train(); D.train()
    losses = [float(train_step(G, D, opt, s, msg, grad_sync=sync)["total"]) for _ in range(2)]
    torch.cuda.synchronize()
    torch.save({"flat": opt.flat.cpu(), "losses": losses, "bucket_bytes": 4 * opt.grad.numel()}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
