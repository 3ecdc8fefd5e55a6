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


def main():
    out_dir, n_total, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")      # no hostname lookup (it may not resolve on the box)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    import awm_amd
    from awm_amd import distributed as wmd
    from oracle import recipes as R, wm_oracle as O
    awm_amd.lib.load()
    gsd, dsd = R.reference_layout_init()
    R.perturb_bn_(gsd, R.BN_SEED_G); R.perturb_bn_(dsd, R.BN_SEED_D)
    G, D = awm_amd.Generator(16), awm_amd.Detector(16)
    G.load_state_dict(gsd); D.load_state_dict(dsd)
    G.to(dev); D.to(dev)
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
    # phase 1
    G.eval(); D.eval()
    opt.zero_grad()
    total, _ = awm_amd.forward_losses(G, D, s, msg)
    total.backward()
    opt.finish_backward()
    sync()
    torch.cuda.synchronize()
    if rank == 0:
        torch.save(opt.grad.cpu(), os.path.join(out_dir, "avg_grad.pt"))
    # phase 2
    G.train(); D.train()
    losses = [float(awm_amd.train_step(G, D, opt, s, msg, grad_sync=sync)["total"]) for _ in range(2)]
    torch.cuda.synchronize()
    torch.save({"flat": opt.flat.cpu(), "losses": losses}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
