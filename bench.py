#!/usr/bin/env python
"""bench.py -- clips/s of the watermark embed+detect train step (BASELINE.json metric) on N MI355X.

A "step" is one pass of the reference's hot loop body (py/main16.py:242-278) over one synthetic batch:
Generator -> fir/clamp/rms -> Detector on [watermarked; clean] -> {l1, mel, loud, loc, bce, hf} -> backward
-> Adam update, fp32, inputs resident in HBM.  N=1 workload = BASELINE.json configs[2] (B=256 train step);
N>1 = configs[3]: the same per-GPU batch on every rank (weak scaling) + the gradient all-reduce over RCCL
(distributed.GradSync: Detector span from post-accumulate hooks while backward still runs, the rest after).
Prints ONE JSON line on rank 0.  With no flags at N=1 the line also carries `other_configs`: short measurements
of BASELINE configs[1] (B=64 eval-mode forward) and configs[4] (main14b_2, hidden 256, B=128 train step), taken
after the headline region.

  python bench.py --gpus 1 --steps 5 --warmup 2
  python bench.py --gpus N ...            (starts its own N ranks)   or
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
  python bench.py --model main14b_2       (configs[4] as the headline line, with its own roofline / cpu_baseline)
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
PMC_MAIN16 = "r02_pmc_fetch_write_b256.json"          # profiles/: per-kernel HBM bytes, made by profiles/summarize_pmc.py
PMC_MAIN14B2 = "r02_pmc_fetch_write_main14b2_b128.json"


def synthetic_batch(batch, rank, dev):
    """s = clamp(0.1*randn, +-0.99), seed 1234+rank; message uniform in [0, 65536), seed 4321+rank"""
    g = torch.Generator().manual_seed(1234 + rank)
    s = (0.1 * torch.randn(batch, 1, T, generator=g)).clamp_(-0.99, 0.99)
    g2 = torch.Generator().manual_seed(4321 + rank)
    msg = torch.randint(0, 65536, (batch,), generator=g2, dtype=torch.int64)
    return s.to(dev), msg.to(dev)


class LaunchTimer:
    """HIP-event bracket around every launch of one C-ABI entry point that matches `pred`, recorded on the stream the
    kernel is launched on (torch's current stream == the stream handed to the launcher).  `work(args)` returns the
    ALGORITHMIC (flops, bytes) of that launch."""

    def __init__(self, lib, name, pred, work):
        self.events, self.on, self.lib, self.name = [], False, lib, name
        self.orig = getattr(lib, name)

        def wrapped(*a):
            if self.on and pred(a):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                self.orig(*a)
                e1.record()
                self.events.append((e0, e1) + tuple(work(a)))
            else:
                self.orig(*a)
        setattr(lib, name, wrapped)

    def restore(self):
        setattr(self.lib, self.name, self.orig)

    def totals(self):
        """(launches, total ms, total flops, total bytes)"""
        ms = sum(a.elapsed_time(b) for a, b, _, _ in self.events)
        return len(self.events), ms, sum(e[2] for e in self.events), sum(e[3] for e in self.events)


def pmc_traffic(fname, prefixes):
    """mean HBM bytes per launch of the dominant kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
    in separate runs, gfx950 FETCH x2 correction; profiles/summarize_pmc.py) -- None when the file is absent"""
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", fname)))["kernels"]
        ks = [v["hbm_bytes_per_launch_corrected"] for k, v in pm.items() if k.startswith(tuple(prefixes))]
        return sum(ks) / len(ks) if ks else None
    except Exception:
        return None


def cpu_baseline_main16(sample_batch=2, steps=1):
    """The oracle (CPU restatement, kind "port") timed on this box's host cores on a bounded sample of the same workload:
    B=2 train steps (fwd + bwd + Adam), 1 warm-up + `steps` timed (about 15-20 s of CPU work in all)."""
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


def cpu_baseline_main14b2(G, D, sample_batch=2, steps=1):
    """same for configs[4]: oracle/wm_oracle_14b2.py step (hidden 256) on the host cores, weights = the modules' own"""
    from oracle import wm_oracle as O
    from oracle import wm_oracle_14b2 as O2
    nthreads = torch.get_num_threads()
    gsd = {k: v.detach().cpu().clone().requires_grad_(v.is_floating_point()) for k, v in G.state_dict().items()}
    dsd = {k: v.detach().cpu().clone().requires_grad_(v.is_floating_point()) for k, v in D.state_dict().items()}
    opt = torch.optim.Adam([v for v in list(gsd.values()) + list(dsd.values()) if v.requires_grad], lr=1e-3)
    s = O.synthetic_clips(sample_batch, seed=1234)
    msg = O.synthetic_messages(sample_batch, seed=4321)

    def one():
        opt.zero_grad()
        total, _ = O2.step_losses(gsd, dsd, s, msg)
        total.backward()
        opt.step()
    one()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    dt = (time.perf_counter() - t0) / steps
    return {"value": sample_batch / dt, "unit": "clips/s", "cores": nthreads, "kind": "port",
            "sample": f"oracle/wm_oracle_14b2.py train step (hidden 256, fwd+bwd+Adam), B={sample_batch}, 1 warm-up + {steps} timed "
                      f"steps, {dt:.2f} s/step, torch CPU fp32 with {nthreads} threads"}


def build_workload(model, mode, batch, rank, world, dev, torch_adam=False, force_sync=False):
    """modules + optimizer + step closure + the dominant-kernel timer of one workload"""
    import awm_amd
    from awm_amd import distributed as wmd
    from awm_amd import ops as _ops
    torch.manual_seed(42)                               # weights: PyTorch default init under manual_seed(42)
    if model == "main14b_2":
        from awm_amd import main14b_2 as M14
        G, D = M14.Generator(hidden_dim=256), M14.Detector()
        step_fn = M14.train_step
    else:
        G, D = awm_amd.Generator(16), awm_amd.Detector(16)
        step_fn = awm_amd.train_step
    G.to(dev); D.to(dev)
    (G.train(), D.train()) if mode == "train" else (G.eval(), D.eval())
    wmd.broadcast_parameters([G, D])
    sync = None
    if mode == "train":
        if torch_adam:
            opt = torch.optim.Adam(list(G.parameters()) + list(D.parameters()), lr=1e-3)
            if world > 1:
                sync = lambda: wmd.allreduce_gradients(list(G.parameters()) + list(D.parameters()))   # noqa: E731
        else:
            # WM_OVERLAP_WGRAD=1: weight-gradient GEMMs on a side stream (co-resident with the LSTM recurrences), A/B knob
            opt = awm_amd.FlatAdam([G, D], lr=1e-3, overlap_wgrad=os.environ.get("WM_OVERLAP_WGRAD") == "1")
            gsync = wmd.GradSync(opt, early_modules=[D], force=force_sync)
            sync = gsync if (world > 1 or force_sync) else None
    s, msg = synthetic_batch(batch, rank, dev)
    if mode == "fwd":
        def step():                                     # evaluate_model's forward, py/main16.py:383-403
            return awm_amd.eval_forward(G, D, s, msg)
    else:
        def step():
            return step_fn(G, D, opt, s, msg, grad_sync=sync)

    bf_mode = _ops.conv_bf16x6()
    if model == "main14b_2":
        # dominant kernel: gconv2_kernel<2,2,2> -- the 128 x 128-tile build of the generic implicit-GEMM convolution
        # (wm_gconv with > 64 output rows, > 64 output positions, 16-byte-aligned weight rows): forward convolutions,
        # transposed convolutions and every data gradient of the wide layers.
        # args: x wp bias vec res y NB Cin Lin K S P Mtot Nout st shp Cout Lout act stream
        timer = LaunchTimer(awm_amd.lib, "wm_gconv", lambda a: a[12] > 64 and a[13] > 64 and a[12] % 4 == 0,
                            lambda a: (2.0 * a[6] * a[7] * a[9] * a[12] * a[13], 4.0 * a[6] * (a[7] * a[8] + a[16] * a[17])))
    elif bf_mode and mode == "fwd" and _ops._CONV["one_launch_eval"]:
        # inference: the ResBlock is ONE launch (two 64->64 k3 convolutions back to back; x in, out out)
        # args: x w1pb w2pb b1 sc1 sh1 b2 sc2 sh2 y B T stream
        timer = LaunchTimer(awm_amd.lib, "wm_resblock_eval_bf", lambda a: True,
                            lambda a: (2.0 * 2.0 * 64 * 64 * 3 * a[11] * a[10], 2.0 * 64 * a[11] * 4 * a[10]))
    elif bf_mode:
        # dominant kernel: the 64->64 k3 forward convolution of the ResBlocks (epi = bias): wm_conv64_bf (bf16x6 split build)
        timer = LaunchTimer(awm_amd.lib, "wm_conv64_bf", lambda a: a[15] == 0,
                            lambda a: (2.0 * 64 * 64 * 3 * a[13] * a[12], 2.0 * 64 * a[13] * 4 * a[12]))
    else:
        timer = LaunchTimer(awm_amd.lib, "wm_conv64", lambda a: a[14] == 3 and a[16] == 0,
                            lambda a: (2.0 * 64 * 64 * 3 * a[13] * a[12], 2.0 * 64 * a[13] * 4 * a[12]))
    return G, D, step, timer, bf_mode


def run_workload(model, mode, batch, steps, warmup, rank, world, dev, dist, torch_adam=False, force_sync=False):
    G, D, step, timer, bf_mode = build_workload(model, mode, batch, rank, world, dev, torch_adam, force_sync)
    for _ in range(warmup):
        step()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    timer.on = True
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    fence()
    dt = time.perf_counter() - t0
    timer.on = False
    timer.restore()
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    tot = out["total"] if mode == "train" else out["delta_rms"].mean()
    total_loss = float(tot.detach()) if torch.is_tensor(tot) else float(tot)
    assert total_loss == total_loss, "NaN"
    n, k_ms, flops, byts = timer.totals()
    roofline = None
    if n:
        ach = flops / (k_ms * 1e-3) / 1e12
        if model == "main14b_2":
            peak, note = PEAK_FP32_MFMA_TFLOPS, "fp32 MFMA dense peak (v_mfma_f32_32x32x2_f32)"
            kernel = ("gconv2_kernel<2,2,2> (wm_gconv, 128 x 128 output tiles: implicit-GEMM Conv1d / ConvTranspose1d / data gradients of "
                      "the wide main14b_2 layers on the fp32 matrix cores; achieved = sum of algorithmic FLOPs / sum of launch times)")
            traffic = pmc_traffic(PMC_MAIN14B2, ("gconv2_kernel<2, 2, 2",)) if batch == 128 else None
        else:
            # bf16x6 mode runs on the bf16 matrix pipe (dense peak 2500 TFLOP/s) and spends six piece products per fp32-grade
            # product: its ceiling in ALGORITHMIC flops is 2500/6 = 416.7 TFLOP/s.  Native mode: the fp32 MFMA peak.
            peak = (PEAK_BF16_MFMA_TFLOPS / 6.0) if bf_mode else PEAK_FP32_MFMA_TFLOPS
            note = ("bf16 dense MFMA peak 2500 TFLOP/s / 6 bf16 piece products per fp32-grade product" if bf_mode
                    else "fp32 MFMA dense peak (v_mfma_f32_32x32x2_f32)")
            kernel = ("conv64bf3_kernel forward (wm_conv64_bf: Conv1d(64,64,3)+bias as bf16x6 split products on the bf16 matrix cores, "
                      "fp32 accumulate, fp32-grade error; BN+ReLU fused on load, BN sums in epilogue)") if bf_mode else \
                "conv64_kernel<KW=3> forward (wm_conv64: native fp32 MFMA; BN+ReLU fused on load, BN sums in epilogue)"
            if timer.name == "wm_resblock_eval_bf":
                kernel = ("resblock_eval_kernel (wm_resblock_eval_bf: the inference ResBlock in one launch -- conv1 + BN1 + ReLU, the "
                          "intermediate kept in LDS as bf16x3 pieces, conv2 + BN2 + residual + ReLU; bf16x6 split products)")
            pref = ("conv64bf3_kernel<0, 0", "conv64bf3_kernel<1, 0") if bf_mode else ("conv64_kernel<3, 256",)
            traffic = pmc_traffic(PMC_MAIN16, pref) if batch == 256 else None
        roofline = {"bound": "mfma", "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), "traffic": traffic, "peak_note": note,
                    "achieved_over_fp32_mfma_peak_157TF": round(ach / PEAK_FP32_MFMA_TFLOPS, 4), "kernel": kernel,
                    "avg_launch_ms": round(k_ms / n, 4), "launches_timed": n,
                    "algorithmic_flops_per_launch": flops / n, "algorithmic_bytes_per_launch": byts / n,
                    "hbm_achieved_GBs": round(byts / (k_ms * 1e-3) / 1e9, 1),
                    "hbm_frac_of_8TBs": round(byts / (k_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}
    res = {"ms_per_step": 1e3 * dt / steps, "value": batch * world * steps / dt, "loss": total_loss, "roofline": roofline,
           "bf_mode": bf_mode, "modules": (G, D)}
    return res


def workload_name(model, mode, batch, world):
    if mode == "fwd":
        return (f"main16 eval-mode forward only (Generator -> fir/clamp/rms -> Detector on [watermarked; clean] + evaluate_model "
                f"reductions), B={batch} clips/GPU x {world} GPU (BASELINE configs[1])")
    if model == "main16":
        return (f"main16 train step: Generator+Detector+6 losses fwd-bwd + Adam, B={batch} clips/GPU x {world} GPU, "
                f"1-s @ 16 kHz, message_bits=16 (BASELINE configs[{2 if world == 1 else 3}])")
    return (f"main14b_2 deep-residual train step (hidden_dim=256, 2-layer LSTM): Generator+Detector+5 losses fwd-bwd + Adam, "
            f"B={batch} clips/GPU x {world} GPU (BASELINE configs[4])")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="clips per GPU (default: 256 for main16 = BASELINE configs[2]/[3], "
                                                            "128 for main14b_2 = configs[4], 64 for --mode fwd = configs[1])")
    ap.add_argument("--model", choices=["main16", "main14b_2"], default="main16",
                    help="main16: BASELINE configs[2]/[3] (default, the headline metric); main14b_2: configs[4] "
                         "(deep-residual variant, hidden_dim=256)")
    ap.add_argument("--mode", choices=["train", "fwd"], default="train",
                    help="train: fwd+bwd+Adam (the BASELINE metric, default); fwd: eval-mode forward only "
                         "(BASELINE configs[1]) -- reported with its own metric name")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the short configs[1] / configs[4] measurements of the default run")
    ap.add_argument("--torch-adam", action="store_true", help="use torch.optim.Adam instead of the fused flat Adam")
    args = ap.parse_args()
    if args.batch is None:
        args.batch = 64 if args.mode == "fwd" else (128 if args.model == "main14b_2" else 256)
    if args.mode == "fwd" and args.model != "main16":
        raise SystemExit("--mode fwd is wired for main16")

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
    # WM_DIST_BACKEND=gloo replaces RCCL, so the N>1 code path can be exercised on a one-GPU box;
    # WM_FORCE_SYNC=1 runs the gradient exchange on a one-rank RCCL communicator
    if os.environ.get("WM_BENCH_SHARE_GPU") == "1":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    force_sync = os.environ.get("WM_FORCE_SYNC") == "1"
    if world > 1 or force_sync:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        backend = os.environ.get("WM_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            if backend == "gloo":
                os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")      # single-node rehearsal: no hostname lookup
            dist.init_process_group(backend, rank=rank, world_size=world)

    import awm_amd
    awm_amd.lib.load()                                  # fail loudly if the HIP library is missing

    r = run_workload(args.model, args.mode, args.batch, args.steps, args.warmup, rank, world, dev, dist, args.torch_adam, force_sync)
    if rank == 0:
        bf_mode = r["bf_mode"]
        line = {"metric": "1-s@16kHz clips/sec (gen+det+loss fwd-bwd)" if args.mode == "train" else
                          "1-s@16kHz clips/sec (gen+det forward only, eval mode)",
                "value": round(r["value"], 2), "unit": "clips/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(r["ms_per_step"], 3),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": workload_name(args.model, args.mode, args.batch, world),
                           "batch_per_gpu": args.batch, "global_batch": args.batch * world, "clip_len": T,
                           "optimizer": "torch.optim.Adam" if args.torch_adam else "fused flat Adam (wm_adam_step)",
                           "conv_arithmetic": ("64->64 convs (k3 and k7: fwd, dgrad, wgrad) and the LSTM input projection: bf16x6 split on "
                                               "bf16 MFMA, fp32 accumulate (2.7e-7 vs fp64; native fp32 MFMA 2.5e-7)")
                           if (args.model == "main16" and bf_mode) else "native fp32 MFMA",
                           "parallelism": f"dp{world}" if world > 1 else "single"},
                "loss": round(r["loss"], 6), "roofline": r["roofline"]}
        if world == 1 and not args.no_cpu_baseline and args.mode == "train":
            line["cpu_baseline"] = cpu_baseline_main16() if args.model == "main16" else cpu_baseline_main14b2(*r["modules"])
        default_run = (args.model == "main16" and args.mode == "train" and args.batch == 256 and world == 1 and not args.no_extra
                       and not args.torch_adam)
        del r
        torch.cuda.empty_cache()
        if default_run:
            # the other single-GPU BASELINE configs, measured after the headline region (5 timed steps each): parity-test
            # cases, reported for completeness -- `value` above is configs[2] alone
            extra = {}
            for key, (m, md, b) in (("configs[1]", ("main16", "fwd", 64)), ("configs[4]", ("main14b_2", "train", 128))):
                e = run_workload(m, md, b, 5, 2, rank, world, dev, dist)
                rf = e["roofline"]
                extra[key] = {"workload": workload_name(m, md, b, world), "value": round(e["value"], 2), "unit": "clips/s",
                              "ms_per_step": round(e["ms_per_step"], 3), "steps": 5, "warmup": 2,
                              "roofline": None if rf is None else {k: rf[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel")}}
                del e
                torch.cuda.empty_cache()
            line["other_configs"] = extra
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
