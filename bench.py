#!/usr/bin/env python
"""bench.py -- clips/s of the watermark embed+detect train step (BASELINE.json metric) on N MI355X.

A "step" is one pass of the reference's hot loop body (py/main16.py:242-278) over one synthetic batch:
Generator -> fir/clamp/rms -> Detector on [watermarked; clean] -> {l1, mel, loud, loc, bce, hf} -> backward
-> Adam update, fp32, inputs resident in HBM.  N=1 workload = BASELINE.json configs[2] (B=256 train step);
N>1 = configs[3]: the same per-GPU batch on every rank (weak scaling) + one RCCL all-reduce of the flat
gradient bucket.  Prints ONE JSON line on rank 0.

  python bench.py --gpus 1 --steps 5 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

T = 16000
PEAK_FP32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0      # MI355X_MICROARCH.md: dense bf16 MFMA peak (no sparsity)
PEAK_HBM_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E spec (6.3 TB/s achievable)


def synthetic_batch(batch, rank, dev):
    """s = clamp(0.1*randn, +-0.99), seed 1234+rank; message uniform in [0, 65536), seed 4321+rank"""
    g = torch.Generator().manual_seed(1234 + rank)
    s = (0.1 * torch.randn(batch, 1, T, generator=g)).clamp_(-0.99, 0.99)
    g2 = torch.Generator().manual_seed(4321 + rank)
    msg = torch.randint(0, 65536, (batch,), generator=g2, dtype=torch.int64)
    return s.to(dev), msg.to(dev)


class LaunchTimer:
    """HIP-event bracket around every launch of one C-ABI entry point that matches `pred`, recorded on the
    stream the kernel is launched on (torch's current stream == the stream handed to the launcher)."""

    def __init__(self, lib, name, pred):
        self.events, self.on = [], False
        orig = getattr(lib, name)

        def wrapped(*a):
            if self.on and pred(a):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                orig(*a)
                e1.record()
                self.events.append((e0, e1, a[12]))          # a[12] = clips in this launch
            else:
                orig(*a)
        setattr(lib, name, wrapped)

    def mean_ms(self):
        if not self.events:
            return None
        return sum(a.elapsed_time(b) for a, b, _ in self.events) / len(self.events)

    def mean_clips(self):
        return sum(c for _, _, c in self.events) / len(self.events)


def cpu_baseline(sample_batch=4, steps=1):
    """The oracle (CPU restatement, kind "port") timed on this box's host cores on a bounded sample of the same
    workload: B=4 train steps (fwd + bwd + Adam), 1 warm-up + `steps` timed (about 20 s of CPU work in all, so the
    default run stays a GPU run on the driver's clock)."""
    from oracle import recipes as R
    from oracle import wm_oracle as O
    nthreads = torch.get_num_threads()
    gsd, dsd = R.reference_layout_init()
    params = []
    for sd in (gsd, dsd):
        for k, v in sd.items():
            if v.is_floating_point() and "running" not in k:
                v.requires_grad_(True)
                params.append(v)
    opt = torch.optim.Adam(params, lr=1e-3)
    s = O.synthetic_clips(sample_batch, seed=1234)
    msg = O.synthetic_messages(sample_batch, seed=4321)

    def one():
        opt.zero_grad()
        total, _ = O.step_losses(gsd, dsd, s, msg, training=True, g_stats={}, d_stats={})
        total.backward()
        opt.step()
    one()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    dt = (time.perf_counter() - t0) / steps
    return {"value": sample_batch / dt, "unit": "clips/s", "cores": nthreads, "kind": "port",
            "sample": f"oracle/wm_oracle.py train step (fwd+bwd+Adam), B={sample_batch}, 1 warm-up + {steps} timed steps, "
                      f"{dt:.2f} s/step, torch CPU fp32 with {nthreads} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256, help="clips per GPU (BASELINE configs[2]/[3]: 256)")
    ap.add_argument("--model", choices=["main16", "main14b_2"], default="main16",
                    help="main16: BASELINE configs[2]/[3] (default, the headline metric); main14b_2: configs[4] "
                         "(deep-residual variant, hidden_dim=256, default --batch 128)")
    ap.add_argument("--mode", choices=["train", "fwd"], default="train",
                    help="train: fwd+bwd+Adam (the BASELINE metric, default); fwd: eval-mode forward only "
                         "(BASELINE configs[1]: use --batch 64) -- reported with its own metric name")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--torch-adam", action="store_true", help="use torch.optim.Adam instead of the fused flat Adam")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # bare `python bench.py --gpus N`: start the N rank processes ourselves.  Nothing in THIS process has touched
        # the GPU yet (import torch does not), and the ranks are children -- never a re-exec of a GPU process.
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")).returncode)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    import torch.distributed as dist
    # rehearsal knobs (not used by the driver): WM_BENCH_SHARE_GPU=1 lets several ranks share one card and
    # WM_DIST_BACKEND=gloo replaces RCCL, so the N>1 code path can be exercised on a one-GPU box
    if os.environ.get("WM_BENCH_SHARE_GPU") == "1":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("WM_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import awm_amd
    from awm_amd import distributed as wmd
    awm_amd.lib.load()                                  # fail loudly if the HIP library is missing

    torch.manual_seed(42)                               # weights: PyTorch default init under manual_seed(42)
    if args.model == "main14b_2":
        from awm_amd import main14b_2 as M14
        if args.batch == 256:
            args.batch = 128                            # BASELINE configs[4]: batch 128 per GPU
        G, D = M14.Generator(hidden_dim=256), M14.Detector()
        step_fn = M14.train_step
    else:
        G, D = awm_amd.Generator(16), awm_amd.Detector(16)
        step_fn = awm_amd.train_step
    G.to(dev).train(); D.to(dev).train()
    wmd.broadcast_parameters([G, D])
    if args.torch_adam:
        opt = torch.optim.Adam(list(G.parameters()) + list(D.parameters()), lr=1e-3)
        sync = (lambda: wmd.allreduce_gradients(list(G.parameters()) + list(D.parameters()))) if world > 1 else None
    else:
        opt = awm_amd.FlatAdam([G, D], lr=1e-3)
        # Detector span all-reduced from post-accumulate hooks while the Generator's backward is still running, the rest
        # when backward returns (distributed.GradSync).  WM_FORCE_SYNC=1 runs the same code on a one-rank communicator.
        force = os.environ.get("WM_FORCE_SYNC") == "1"
        if force and world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group(os.environ.get("WM_DIST_BACKEND", "nccl"), rank=0, world_size=1)
        gsync = wmd.GradSync(opt, early_modules=[D], force=force)
        sync = gsync if (world > 1 or force) else None
    s, msg = synthetic_batch(args.batch, rank, dev)

    # dominant kernel: the 64->64 k3 forward convolution of the ResBlocks (epi = bias): wm_conv64_bf (bf16x6 split build,
    # the default) or wm_conv64 with KW = 3 (native fp32 MFMA build, WM_CONV_BF16X6=0)
    from awm_amd import ops as _ops
    bf_mode = _ops.conv_bf16x6()
    if bf_mode:
        timer = LaunchTimer(awm_amd.lib, "wm_conv64_bf", lambda a: a[15] == 0)
    else:
        timer = LaunchTimer(awm_amd.lib, "wm_conv64", lambda a: a[14] == 3 and a[16] == 0)

    def step():
        return step_fn(G, D, opt, s, msg, grad_sync=sync)

    if args.mode == "fwd":
        if args.model != "main16":
            raise SystemExit("--mode fwd is wired for main16")
        G.eval(); D.eval()

        def step():                                     # noqa: F811  (evaluate_model's forward, py/main16.py:383-403)
            return awm_amd.eval_forward(G, D, s, msg)

    for _ in range(args.warmup):
        step()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    timer.on = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    dt = time.perf_counter() - t0
    timer.on = False
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    tot = out["total"] if args.mode == "train" else out["delta_rms"].mean()
    total_loss = float(tot.detach()) if torch.is_tensor(tot) else float(tot)
    assert total_loss == total_loss, "NaN"

    if rank == 0:
        ms_step = 1e3 * dt / args.steps
        value = args.batch * world * args.steps / dt
        k_ms = timer.mean_ms()
        # Generator-side launches carry B clips, Detector-side ones 2B ([watermarked; clean]): use what was launched
        clips_per_launch = timer.mean_clips() if k_ms else 0.0
        flops_launch = 2.0 * 64 * 64 * 3 * T * clips_per_launch
        bytes_launch = 2.0 * 64 * T * 4 * clips_per_launch
        # HBM traffic of the dominant kernel from the committed PMC pass (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
        # separate runs, gfx950 FETCH x2 correction; profiles/r01_pmc_fetch_write_b256_v8.json, profiles/summarize_pmc.py) -- same B=256 workload only
        traffic = None
        try:
            if args.batch == 256:
                pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_fetch_write_b256_v8.json")))["kernels"]
                pref = ("conv64bf3_kernel<0, 0", "conv64bf3_kernel<1, 0") if bf_mode else ("conv64_kernel<3, 256",)
                ks = [v["hbm_bytes_per_launch_corrected"] for k, v in pm.items() if k.startswith(pref)]
                traffic = sum(ks) / len(ks) if ks else None
        except Exception:
            traffic = None
        roofline = None
        if k_ms:
            ach = flops_launch / (k_ms * 1e-3) / 1e12
            # bf16x6 mode runs on the bf16 matrix pipe (dense peak 2500 TFLOP/s) and spends six piece products per
            # fp32-grade product: its ceiling in ALGORITHMIC flops is 2500/6 = 416.7 TFLOP/s (frac = share of the
            # bf16 pipe's time in use).  Native mode is priced against the fp32 MFMA peak.
            peak = (PEAK_BF16_MFMA_TFLOPS / 6.0) if bf_mode else PEAK_FP32_MFMA_TFLOPS
            roofline = {"bound": "mfma", "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                        "frac": round(ach / peak, 4), "traffic": traffic,
                        "peak_note": ("bf16 dense MFMA peak 2500 TFLOP/s / 6 bf16 piece products per fp32-grade product"
                                      if bf_mode else "fp32 MFMA dense peak (v_mfma_f32_32x32x2_f32)"),
                        "achieved_over_fp32_mfma_peak_157TF": round(ach / PEAK_FP32_MFMA_TFLOPS, 4),
                                "kernel": ("conv64bf3_kernel forward (wm_conv64_bf: Conv1d(64,64,3)+bias as bf16x6 split products on the bf16 "
                                   "matrix cores, fp32 accumulate, fp32-grade error; BN+ReLU fused on load, BN sums in epilogue)") if bf_mode else
                                  "conv64_kernel<KW=3> forward (wm_conv64: native fp32 MFMA; BN+ReLU fused on load, BN sums in epilogue)",
                        "bf16_mfma_flops_per_launch": (6.0 * flops_launch) if bf_mode else None,
                        "avg_launch_ms": round(k_ms, 4), "launches_timed": len(timer.events),
                        "algorithmic_flops_per_launch": flops_launch, "algorithmic_bytes_per_launch": bytes_launch,
                        "hbm_achieved_GBs": round(bytes_launch / (k_ms * 1e-3) / 1e9, 1),
                        "hbm_frac_of_8TBs": round(bytes_launch / (k_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}
        if args.model == "main14b_2":
            roofline = None        # the generic-shape kernels of config 5 are correctness-first; no roofline claim yet
        workload = (f"main16 train step: Generator+Detector+6 losses fwd-bwd + Adam, B={args.batch} clips/GPU x {world} GPU, "
                    f"1-s @ 16 kHz, message_bits=16 (BASELINE configs[{2 if world == 1 else 3}])") if args.model == "main16" else \
                   (f"main14b_2 deep-residual train step (hidden_dim=256, 2-layer LSTM): Generator+Detector+5 losses fwd-bwd + Adam, "
                    f"B={args.batch} clips/GPU x {world} GPU (BASELINE configs[4])")
        if args.mode == "fwd":
            workload = (f"main16 eval-mode forward only (Generator -> fir/clamp/rms -> Detector on [watermarked; clean] + evaluate_model "
                        f"reductions), B={args.batch} clips/GPU x {world} GPU (BASELINE configs[1])")
        line = {"metric": "1-s@16kHz clips/sec (gen+det+loss fwd-bwd)" if args.mode == "train" else "1-s@16kHz clips/sec (gen+det forward only, eval mode)", "value": round(value, 2), "unit": "clips/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 3),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": workload,
                           "batch_per_gpu": args.batch, "global_batch": args.batch * world, "clip_len": T,
                           "optimizer": "torch.optim.Adam" if args.torch_adam else "fused flat Adam (wm_adam_step)",
                           "conv_arithmetic": "64->64 convs (k3 and k7: fwd, dgrad, wgrad) and the LSTM input projection: bf16x6 split on bf16 MFMA, fp32 accumulate (2.7e-7 vs fp64; native fp32 MFMA 2.5e-7)" if (args.model == "main16" and bf_mode) else "native fp32 MFMA",
                           "parallelism": f"dp{world}" if world > 1 else "single"},
                "loss": round(total_loss, 6), "roofline": roofline}
        if world == 1 and not args.no_cpu_baseline and args.model == "main16" and args.mode == "train":
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
