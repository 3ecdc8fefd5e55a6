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
PMC_MAIN16 = "r03_pmc_fetch_write_b256.json"          # profiles/: per-kernel HBM bytes, made by profiles/summarize_pmc.py
PMC_MAIN14B2 = "r03_pmc_fetch_write_main14b2_b128.json"


def kernel_source_sha():
    """SHA-256 over the HIP sources the library is built from -- profiles/summarize_pmc.py stores the same digest in the PMC
    summary, so a `traffic` figure taken on an older build of the kernels is recognised and dropped instead of going stale"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "audio-watermarking-deep-learning-watermarks-for-authenticating-speech_amd", "csrc", "*.h*"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


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
    """(mean HBM bytes per launch of the named kernel, note) from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE in separate runs, gfx950 FETCH x2 correction; profiles/summarize_pmc.py).  (None, why) when the file is absent,
    has no such kernel, or was taken on different kernel sources than the ones this run was built from."""
    try:
        doc = json.load(open(os.path.join(ROOT, "profiles", fname)))
    except Exception:
        return None, f"profiles/{fname} not found"
    if doc.get("kernel_source_sha") != kernel_source_sha():
        return None, f"profiles/{fname} was taken on other kernel sources ({doc.get('kernel_source_sha')}): dropped, not reported stale"
    ks = [v["hbm_bytes_per_launch_corrected"] for k, v in doc["kernels"].items() if k.startswith(tuple(prefixes))]
    if not ks:
        return None, f"no kernel named {prefixes} in profiles/{fname}"
    return sum(ks) / len(ks), f"profiles/{fname}, mean over kernels {list(prefixes)}"


def host_cpus():
    """(CPUs this process may use, CPU model): the scheduler affinity mask capped by the cgroup quota -- a one-GPU box hands a
    share of the host's cores to the job, and a thread pool sized by the machine's core count only oversubscribes it"""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    try:
        import psutil
        n = max(1, min(n, psutil.cpu_count(logical=False) or n))
    except Exception:
        pass
    model = "unknown CPU"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    return n, model


def _cpu_leg(one_small, one_full, sample_batch, budget_s):
    """BASELINE.md section 3's plan within a time budget: for n in {8 threads, every core the job may use}: one warm-up step (on a
    2-clip batch: thread pool, allocator, oneDNN primitives), then B=`sample_batch` steps until the budget share is used (>= 1)."""
    avail, model = host_cpus()
    runs = []
    counts = sorted({min(8, avail), avail})
    prev = torch.get_num_threads()
    try:
        for n in counts:
            torch.set_num_threads(n)
            one_small()
            t0, k = time.perf_counter(), 0
            while k == 0 or (time.perf_counter() - t0 < budget_s / len(counts) and k < 5):
                one_full()
                k += 1
            dt = (time.perf_counter() - t0) / k
            runs.append({"threads": n, "clips_per_s": round(sample_batch / dt, 3), "s_per_step": round(dt, 2), "timed_steps": k})
    finally:
        torch.set_num_threads(prev)
    best = max(runs, key=lambda r: r["clips_per_s"])
    return runs, best, avail, model


def cpu_baseline_main16(sample_batch=16, budget_s=24.0):
    """The oracle (CPU restatement, kind "port") timed on this box's host cores on a bounded sample of the same workload:
    train steps (fwd + bwd + Adam) at B=16 -- the reference's own BATCH_SIZE (py/main16.py:32) -- with 8 threads and with every core
    the job may use; `value` is the faster of the two, both are listed under `runs`."""
    from oracle import recipes as R
    from oracle import wm_oracle as O
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

    def one(b):
        opt.zero_grad()
        total, _ = O.step_losses(gsd, dsd, s[:b], msg[:b], training=True, g_stats={}, d_stats={})
        total.backward()
        opt.step()
    runs, best, avail, model = _cpu_leg(lambda: one(2), lambda: one(sample_batch), sample_batch, budget_s)
    return {"value": best["clips_per_s"], "unit": "clips/s", "cores": best["threads"], "kind": "port", "runs": runs,
            "cpu_model": model, "cpus_available_to_job": avail,
            "sample": f"oracle/wm_oracle.py train step (fwd+bwd+Adam), B={sample_batch}, torch CPU fp32 on {model}; per thread count one "
                      f"2-clip warm-up step then B={sample_batch} steps within {budget_s:.0f} s in all: "
                      + "; ".join(f"{r['threads']} threads {r['s_per_step']} s/step x {r['timed_steps']}" for r in runs)}


def cpu_baseline_main14b2(G, D, sample_batch=16, budget_s=24.0):
    """same for configs[4]: oracle/wm_oracle_14b2.py step (hidden 256) on the host cores, weights = the modules' own"""
    from oracle import wm_oracle as O
    from oracle import wm_oracle_14b2 as O2
    gsd = {k: v.detach().cpu().clone().requires_grad_(v.is_floating_point()) for k, v in G.state_dict().items()}
    dsd = {k: v.detach().cpu().clone().requires_grad_(v.is_floating_point()) for k, v in D.state_dict().items()}
    opt = torch.optim.Adam([v for v in list(gsd.values()) + list(dsd.values()) if v.requires_grad], lr=1e-3)
    s = O.synthetic_clips(sample_batch, seed=1234)
    msg = O.synthetic_messages(sample_batch, seed=4321)

    def one(b):
        opt.zero_grad()
        total, _ = O2.step_losses(gsd, dsd, s[:b], msg[:b])
        total.backward()
        opt.step()
    runs, best, avail, model = _cpu_leg(lambda: one(2), lambda: one(sample_batch), sample_batch, budget_s)
    return {"value": best["clips_per_s"], "unit": "clips/s", "cores": best["threads"], "kind": "port", "runs": runs,
            "cpu_model": model, "cpus_available_to_job": avail,
            "sample": f"oracle/wm_oracle_14b2.py train step (hidden 256, fwd+bwd+Adam), B={sample_batch}, torch CPU fp32 on {model}; per "
                      f"thread count one 2-clip warm-up step then B={sample_batch} steps within {budget_s:.0f} s in all: "
                      + "; ".join(f"{r['threads']} threads {r['s_per_step']} s/step x {r['timed_steps']}" for r in runs)}


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
    return G, D, step, kernel_timers(awm_amd.lib, model, mode, bf_mode, batch), bf_mode


BF16X6_PEAK = PEAK_BF16_MFMA_TFLOPS / 6.0
BF16X6_NOTE = "bf16 dense MFMA peak 2500 TFLOP/s / 6 bf16 piece products per fp32-grade product"
FP32_NOTE = "fp32 MFMA dense peak (v_mfma_f32_32x32x2_f32)"


def kernel_timers(lib, model, mode, bf_mode, batch):
    """HIP-event timers on the C-ABI entry points whose kernels carry the workload, the DOMINANT one first (share of the step's
    kernel time in the committed rocprofv3 summary under profiles/).  Each entry names the kernel it brackets, its algorithmic
    FLOPs / bytes per launch, the peak it is priced against and where its PMC traffic figure comes from -- the roofline object of
    the JSON line is built from the entry itself, never from strings kept elsewhere."""
    from awm_amd import ops as _ops
    pmc16 = PMC_MAIN16 if batch == 256 else None
    if model == "main14b_2":
        # wm_gconv args: x wp bias vec res y NB Cin Lin K S P Mtot Nout st shp Cout Lout act stream; > 64 output rows and
        # positions with 16-byte-aligned weight rows = the 128 x 128-tile build
        from awm_amd import main14b_2 as _m14
        F16X3_NOTE_ = ("f16 dense MFMA peak 2500 TFLOP/s / 3 f16 piece products per product")
        pmc14 = PMC_MAIN14B2 if batch == 128 else None
        work = lambda a: (2.0 * a[6] * a[7] * a[9] * a[12] * a[13], 4.0 * a[6] * (a[7] * a[8] + a[16] * a[17]))
        fp32 = dict(timer=LaunchTimer(lib, "wm_gconv", lambda a: a[12] > 64 and a[13] > 64 and a[12] % 4 == 0, work),
                    kernel="gconv2_kernel<2,2,2> (wm_gconv, 128 x 128 output tiles: implicit-GEMM Conv1d / ConvTranspose1d / data "
                           "gradients of the wide main14b_2 layers on the fp32 matrix cores; achieved = sum of algorithmic FLOPs / "
                           "sum of launch times)",
                    peak=PEAK_FP32_MFMA_TFLOPS, note=FP32_NOTE, pmc=pmc14, pmc_prefixes=("gconv2_kernel<2, 2, 2",))
        # wm_gwgrad args: A Bx G dbias slab NB Ca Cb La Lb K P ...: G[a][b][k] = sum_{nb,t} A[nb][a][t] Bx[nb][b][t + k - P]
        wg = dict(timer=LaunchTimer(lib, "wm_gwgrad", lambda a: True,
                                    lambda a: (2.0 * a[5] * a[6] * a[7] * a[10] * a[8], 4.0 * a[5] * (a[6] * a[8] + a[7] * a[9]))),
                  kernel="gwgrad2_kernel (wm_gwgrad: every weight gradient of the variant, deterministic split-K GEMM on the fp32 matrix "
                         "cores; all launches; since the convolutions moved to the f16 split the largest family of the step)",
                  peak=PEAK_FP32_MFMA_TFLOPS, note=FP32_NOTE, pmc=pmc14, pmc_prefixes=("gwgrad2_kernel<",))
        if not (_m14._GCONV["f16x3"] and bf_mode):
            return [fp32, wg]
        # wm_gconv_h: the same argument positions (wph in place of wp)
        h = dict(timer=LaunchTimer(lib, "wm_gconv_h", lambda a: a[12] > 64 and a[13] > 64, work),
                 kernel="gconvh_kernel<2,2,2,*> (wm_gconv_h, 128 x 128 output tiles: implicit-GEMM Conv1d / ConvTranspose1d / data gradients "
                        "of the wide main14b_2 layers as f16 two-piece split products on v_mfma_f32_32x32x16_f16, fp32 accumulate; achieved = "
                        "sum of algorithmic FLOPs / sum of launch times)",
                 peak=PEAK_BF16_MFMA_TFLOPS / 3.0, note=F16X3_NOTE_, pmc=pmc14, pmc_prefixes=("gconvh_kernel<2, 2, 2",))
        return [h, wg]
    if not bf_mode:
        return [dict(timer=LaunchTimer(lib, "wm_conv64", lambda a: a[14] == 3 and a[16] == 0,
                                       lambda a: (2.0 * 64 * 64 * 3 * a[13] * a[12], 2.0 * 64 * a[13] * 4 * a[12])),
                     kernel="conv64_kernel<KW=3> forward (wm_conv64: native fp32 MFMA; BN+ReLU fused on load, BN sums in epilogue)",
                     peak=PEAK_FP32_MFMA_TFLOPS, note=FP32_NOTE, pmc=pmc16, pmc_prefixes=("conv64_kernel<3, 256",))]
    fh = _ops._CONV["fwd_f16x3"] and _ops._CONV["schedule"] == 2     # forward arithmetic: f16 two-piece split | bf16x6
    F16X3_NOTE = ("f16 dense MFMA peak 2500 TFLOP/s / 3 f16 piece products per product (the bf16x6 build of rounds 1-2 was priced "
                  "against 2500 / 6 = 416.7)")
    fwd = dict(timer=LaunchTimer(lib, "wm_conv64_bf", lambda a: a[15] == 0,
                                 lambda a: (2.0 * 64 * 64 * 3 * a[13] * a[12], 2.0 * 64 * a[13] * 4 * a[12])),
               kernel="conv64bf3_kernel forward (wm_conv64_bf, epilogue = bias: Conv1d(64,64,3) as " +
                      ("f16 two-piece split, three piece products per product on v_mfma_f32_32x32x16_f16" if fh else
                       "bf16x6 split products on the bf16 matrix cores") +
                      ", fp32 accumulate, fp32-grade error; BN+ReLU fused on load, BN sums in the epilogue)",
               peak=(PEAK_BF16_MFMA_TFLOPS / 3.0) if fh else BF16X6_PEAK, note=F16X3_NOTE if fh else BF16X6_NOTE, pmc=pmc16,
               pmc_prefixes=("conv64bf3_kernel<0, 0", "conv64bf3_kernel<1, 0"))
    if mode == "fwd":
        if not _ops._CONV["one_launch_eval"]:
            return [fwd]
        # args: x w1pb w2pb b1 sc1 sh1 b2 sc2 sh2 y B T stream
        return [dict(timer=LaunchTimer(lib, "wm_resblock_eval_bf", lambda a: True,
                                       lambda a: (2.0 * 2.0 * 64 * 64 * 3 * a[11] * a[10], 2.0 * 64 * a[11] * 4 * a[10])),
                     kernel="resblock_eval_kernel (wm_resblock_eval_bf: the inference ResBlock in one launch -- conv1 + BN1 + ReLU, the "
                            "intermediate kept in LDS as split pieces, conv2 + BN2 + residual + ReLU; " +
                            ("f16 two-piece split, three piece products per product)" if _ops._CONV["eval_f16x3"] else "bf16x6 split products)"),
                     peak=(PEAK_BF16_MFMA_TFLOPS / 3.0) if _ops._CONV["eval_f16x3"] else BF16X6_PEAK,
                     note=F16X3_NOTE if _ops._CONV["eval_f16x3"] else BF16X6_NOTE, pmc=None, pmc_prefixes=())]
    if not _ops._CONV["fused_bwd"]:
        return [fwd]
    # ResBlock backward: data gradient + weight gradient of one Conv1d(64,64,3) in ONE launch = two GEMMs of 2*64*64*3*T*B flops.
    # args: dz y A Bc C wpb xin sc sh e1 ea eb dx stats wpart dw db B T pro epi accumulate gmask stream; epi 1 = conv2 pair (reads
    # dz2 y2 y1, writes dz1: 4 frames), epi 2 = conv1 pair (reads dz1 y1 x dz2, writes dx: 5 frames), epi 8 = conv1 pair that also
    # does the previous block's ReLU backward + BatchNorm sums (one more frame read: 6)
    h = _ops._CONV["bwd_f16x3"]     # backward arithmetic: f16 two-piece split (three products per product) | bf16x6 (six)
    dw = dict(timer=LaunchTimer(lib, "wm_dwgrad64_bf", lambda a: True,
                                lambda a: (2.0 * 2.0 * 64 * 64 * 3 * a[18] * a[17], {1: 4.0, 2: 5.0, 8: 6.0}[a[20]] * 64 * a[18] * 4 * a[17])),
              kernel="dwgrad64bf_kernel<1,1,..> + <2,0,..> + <8,0,..> (wm_dwgrad64_bf: data gradient AND weight gradient of a ResBlock "
                     "Conv1d(64,64,3) in one launch -- BN-backward rebuilt and the ReLU mask applied on load, ReLU mask / BN sums / residual add "
                     "in the epilogue; " + ("f16 two-piece split, three piece products per product on v_mfma_f32_32x32x16_f16" if h else
                                            "bf16x6 split products") + ", fp32 accumulate; the largest share of the step's kernel time)",
              peak=(PEAK_BF16_MFMA_TFLOPS / 3.0) if h else BF16X6_PEAK,
              note=F16X3_NOTE if h else BF16X6_NOTE,
              pmc=pmc16, pmc_prefixes=("dwgrad64bf_kernel<",))
    dw["limited_by"] = ("package power: launched back to back this kernel holds the package at its 1400 W limit with a lowered clock "
                        "(profiles/r03_power_clock.txt); in-kernel stamps: the same 5.75 K cycles per tile at 1.01 GHz on dense operands and "
                        "1.26 GHz on all-zero ones (profiles/r03_dwgrad_stamps.txt) -- neither the HBM nor the MFMA fraction can reach 1")
    return [dw, fwd]


def roofline_of(entry):
    """the `roofline` object of one timed kernel family (None when it was never launched in the timed region)"""
    n, k_ms, flops, byts = entry["timer"].totals()
    if not n:
        return None
    ach = flops / (k_ms * 1e-3) / 1e12
    gbs = byts / (k_ms * 1e-3) / 1e9
    traffic, tnote = pmc_traffic(entry["pmc"], entry["pmc_prefixes"]) if entry["pmc"] else (None, "no PMC pass exists for this batch size")
    f_mfma, f_hbm = ach / entry["peak"], gbs / PEAK_HBM_GBS
    # the bound that is named is the resource the kernel uses the larger share of; the other pair of numbers stays in the object
    if f_hbm > f_mfma:
        head = {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(f_hbm, 4)}
    else:
        head = {"bound": "mfma", "achieved": round(ach, 2), "peak": round(entry["peak"], 1), "unit": "TFLOP/s", "frac": round(f_mfma, 4)}
    head.update({"traffic": traffic, "traffic_source": tnote, "kernel": entry["kernel"],
                 "avg_launch_ms": round(k_ms / n, 4), "launches_timed": n,
                 "algorithmic_flops_per_launch": flops / n, "algorithmic_bytes_per_launch": byts / n,
                 "hbm_achieved_GBs": round(gbs, 1), "hbm_frac_of_8TBs": round(f_hbm, 4),
                 "mfma_achieved_TFLOPs": round(ach, 2), "mfma_peak_TFLOPs": round(entry["peak"], 1), "mfma_frac": round(f_mfma, 4),
                 "mfma_peak_note": entry["note"],
                 "achieved_over_fp32_mfma_peak_157TF": round(ach / PEAK_FP32_MFMA_TFLOPS, 4),
                 "achieved_over_bf16x6_ceiling_416.7TF": round(ach / BF16X6_PEAK, 4)})
    if entry.get("limited_by"):
        head["limited_by"] = entry["limited_by"]
    return head


def run_workload(model, mode, batch, steps, warmup, rank, world, dev, dist, torch_adam=False, force_sync=False):
    G, D, step, timers, bf_mode = build_workload(model, mode, batch, rank, world, dev, torch_adam, force_sync)
    for _ in range(warmup):
        step()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # no cyclic-GC pass inside the timed region: a generation-2 collection of this process (torch's import alone leaves a few million
    # tracked objects) takes 85-300 ms -- two to six steps -- and lands wherever the allocation counters put it
    # (tests/diag_coldstart.py: one 133-ms step in 300, none with the collector off)
    import gc
    gc.collect()
    gc.disable()
    fence()
    for e in timers:
        e["timer"].on = True
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    fence()
    dt = time.perf_counter() - t0
    gc.enable()
    for e in timers:
        e["timer"].on = False
        e["timer"].restore()
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    tot = out["total"] if mode == "train" else out["delta_rms"].mean()
    total_loss = float(tot.detach()) if torch.is_tensor(tot) else float(tot)
    assert total_loss == total_loss, "NaN"
    roofs = [r for r in (roofline_of(e) for e in timers) if r is not None]
    res = {"ms_per_step": 1e3 * dt / steps, "value": batch * world * steps / dt, "loss": total_loss,
           "roofline": roofs[0] if roofs else None, "roofline_other_kernels": roofs[1:], "bf_mode": bf_mode, "modules": (G, D)}
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
                           "conv_arithmetic": ("64->64 convs (k3 ResBlock forward + fused data / weight gradient, k7 ConvTranspose forward / "
                                               "dgrad / wgrad, inference ResBlock): f16 two-piece split, three piece products per product on "
                                               "v_mfma_f32_32x32x16_f16, fp32 accumulate (error vs fp64 within 3x of bf16x6's 2.7e-7; native fp32 "
                                               "MFMA 2.5e-7); LSTM projection / dx / weight gradients and the ragged-length fallbacks: bf16x6 "
                                               "split on bf16 MFMA")
                           if (args.model == "main16" and bf_mode) else "native fp32 MFMA",
                           "parallelism": f"dp{world}" if world > 1 else "single"},
                "loss": round(r["loss"], 6), "roofline": r["roofline"]}
        if r["roofline_other_kernels"]:
            line["roofline_other_kernels"] = r["roofline_other_kernels"]
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
