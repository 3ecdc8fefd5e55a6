"""Multi-GPU rehearsal on ONE GPU: two fresh child processes run the real HIP modules + FlatAdam + GradSync as
world_size-2 ranks sharing the card over gloo (tests/dist_rehearsal.py), and bench.py's own N>1 launch path is driven
the same way.  What cannot be covered here is RCCL over xGMI itself (needs >= 2 GPUs; the driver's scaling run)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

from oracle import recipes as R
from oracle import wm_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _spawn_ranks(script_args, world, extra_env=None, timeout=600):
    """start `world` rank processes (children of this pytest process; nothing is re-exec'ed) and wait for all"""
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable] + script_args, env=env, cwd=ROOT, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o[-3000:]}"
    return outs


@pytest.mark.timeout(900)
def test_two_ranks_share_one_gpu_hip_path(tmp_path):
    import awm_amd
    dev = torch.device("cuda:0")
    n_total, T = 4, 2048
    _spawn_ranks([os.path.join(ROOT, "tests", "dist_rehearsal.py"), str(tmp_path), str(n_total), str(T)], world=2)
    r0, r1 = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    # (ii) replicas identical after two exchanged steps (bit-exact: same averaged gradient, same Adam arithmetic)
    assert torch.equal(r0["flat"], r1["flat"]), float((r0["flat"] - r1["flat"]).abs().max())
    assert all(abs(a) < 1e6 for a in r0["losses"] + r1["losses"])
    # (iii) rank-averaged flat gradient == single-process gradient of the concatenated batch (eval-mode BatchNorm)
    gsd, dsd = R.reference_layout_init()
    R.perturb_bn_(gsd, R.BN_SEED_G); R.perturb_bn_(dsd, R.BN_SEED_D)
    G, D = awm_amd.Generator(16), awm_amd.Detector(16)
    G.load_state_dict(gsd); D.load_state_dict(dsd)
    G.to(dev).eval(); D.to(dev).eval()
    opt = awm_amd.FlatAdam([G, D], lr=1e-3)
    opt.zero_grad()
    s = O.synthetic_clips(n_total, seed=41, T=T).to(dev)
    msg = O.synthetic_messages(n_total, seed=42).to(dev)
    total, _ = awm_amd.forward_losses(G, D, s, msg)
    total.backward()
    opt.finish_backward()
    big = opt.grad.cpu()
    avg = torch.load(tmp_path / "avg_grad.pt")
    # per-parameter: fp32 summation order differs between one 4-clip batch and two 2-clip shards
    for (off, k), p in zip(opt._spans, opt.params):
        a, b = avg[off:off + k], big[off:off + k]
        scale = float(b.abs().max())
        if scale == 0.0:
            assert float(a.abs().max()) == 0.0
            continue
        assert float((a - b).abs().max()) <= 2e-3 * scale + 1e-7, (off, k, float((a - b).abs().max()), scale)


@pytest.mark.timeout(900)
def test_two_ranks_share_one_gpu_main14b2(tmp_path):
    """BASELINE configs[4]'s exchange (py/main14b_2.py:300-352 step, 24 893 874 parameters = 99.6 MB bucket): two ranks with the
    real HIP modules; replicas bit-identical after two exchanged steps, rank-averaged gradient = big-batch gradient."""
    import awm_amd
    from awm_amd import main14b_2 as M14
    dev = torch.device("cuda:0")
    n_total, T = 4, 6400
    _spawn_ranks([os.path.join(ROOT, "tests", "dist_rehearsal.py"), str(tmp_path), str(n_total), str(T), "main14b_2"], world=2)
    r0, r1 = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    assert r0["bucket_bytes"] == 4 * 24893874
    assert torch.equal(r0["flat"], r1["flat"]), float((r0["flat"] - r1["flat"]).abs().max())
    assert all(abs(a) < 1e6 for a in r0["losses"] + r1["losses"])
    torch.manual_seed(42)
    G, D = M14.Generator(hidden_dim=256).to(dev).train(), M14.Detector().to(dev).train()
    opt = awm_amd.FlatAdam([G, D], lr=1e-3)
    opt.zero_grad()
    s = O.synthetic_clips(n_total, seed=41, T=T).to(dev)
    msg = O.synthetic_messages(n_total, seed=42).to(dev)
    M14.forward_losses(G, D, s, msg)[0].backward()
    opt.finish_backward()
    big, avg = opt.grad.cpu(), torch.load(tmp_path / "avg_grad.pt")
    for (off, k), p in zip(opt._spans, opt.params):
        a, b = avg[off:off + k], big[off:off + k]
        scale = float(b.abs().max())
        if scale == 0.0:
            assert float(a.abs().max()) == 0.0
            continue
        assert float((a - b).abs().max()) <= 3e-3 * scale + 1e-7, (off, k, float((a - b).abs().max()), scale)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("model", ["main16", "main14b_2"])
def test_gradsync_production_path_on_one_rank_rccl(tmp_path, model):
    """distributed.GradSync's RCCL path (hook-launched async all-reduce of the Detector span on a side stream, join in
    __call__) on a one-rank nccl communicator in a fresh child process: exchanged gradient == plain gradient bit for bit,
    stale-state hazards (a backward without an exchange; a second backward) handled.  tests/dist_rehearsal.py force_sync"""
    T = 2048 if model == "main16" else 3200
    _spawn_ranks([os.path.join(ROOT, "tests", "dist_rehearsal.py"), str(tmp_path), "2", str(T), model, "force_sync"], world=1)
    r = torch.load(tmp_path / "force_sync.pt")
    assert r["first_backward_equals_second"], ("the first backward of the process differs from the second", r["first_vs_second_max_diff"])
    assert r["same"], (r["max_diff"], r["early_span"], r["where"][:12])
    assert r["refused"] and all(abs(a) < 1e6 for a in r["losses"])
    assert r["bucket_bytes"] == 4 * (4383314 if model == "main16" else 24893874)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("model,batch", [("main16", 8), ("main14b_2", 4)])
def test_bench_launches_its_own_ranks(tmp_path, model, batch):
    """`python bench.py --gpus 2` (no torchrun in front, as the driver's bare form) starts two ranks itself; both share
    the one GPU over gloo here.  One JSON line, n_gpus = 2, weak scaling: global batch = 2 x per-GPU batch."""
    env = dict(os.environ, WM_BENCH_SHARE_GPU="1", WM_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("RANK", None); env.pop("WORLD_SIZE", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--model", model,
                        "--batch", str(batch), "--no-cpu-baseline"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["global_batch"] == 2 * batch
    assert line["value"] > 0 and line["loss"] == line["loss"]


@pytest.mark.timeout(900)
def test_bench_line_contract():
    """`python bench.py` on one GPU prints ONE JSON line with the driver's keys, the roofline object of the dominant kernel (timed
    with HIP events inside the run) and the CPU-oracle baseline (small batch here: the contract, not the number)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "8", "--no-extra"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["warmup"] == 1 and line["higher_is_better"] is True
    assert line["unit"] == "clips/s" and line["dtype"] == "f32" and line["data"] == "synthetic" and line["vs_baseline"] is None
    assert "workload" in line["config"] and "model" not in line["config"]
    rf = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] in ("hbm", "mfma") and 0.0 < rf["frac"] < 1.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and "sample" in cb
    assert abs(line["value"] - 8 / (line["ms_per_step"] * 1e-3)) <= 0.01 * line["value"]
