#!/usr/bin/env python
"""Golden fixture G7 for the callers of the hot path, from the REFERENCE itself.

Build container only (needs /root/reference).  AST-extracts the reference's own `evaluate_model` (py/main16.py:369-423)
and `compute_si_snr` (:764-773) together with the hot-path definitions they call, runs them on CPU and stores numbers
only (tests/golden/main16_eval_golden.npz).  `evaluate_unseen_file` / `detect_prob` (:1263-1299, :1575-1596) start with
`torchaudio.load` (torchaudio is absent and is not stubbed): their arithmetic after the load is pinned through the
pieces it is made of -- the Generator / Detector forward (G1, G3), the per-segment mean detection probability (this
file: `seg_mean_prob`, the reference Detector run at B=1 on a zero-padded tail segment) and `compute_si_snr`.

Usage:  python tests/golden/make_golden_eval.py [--ref /root/reference]
"""
from __future__ import annotations

import argparse
import ast
import contextlib
import io
import math
import os
import sys
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from tqdm import tqdm

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import wm_oracle as O  # noqa: E402
from oracle import recipes as R    # noqa: E402

WANTED = ["fir_lowpass", "clamp_peak", "limit_rms", "ResBlock", "Generator", "Detector", "evaluate_model", "compute_si_snr"]
EVAL_SEED = 2024


def extract(ref_root):
    tree = ast.parse(open(os.path.join(ref_root, "py", "main16.py")).read())
    seen, nodes = set(), []
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in WANTED and node.name not in seen:
            seen.add(node.name)
            nodes.append(node)
    assert not set(WANTED) - seen
    ns = {"torch": torch, "nn": nn, "F": F, "math": math, "np": np, "tqdm": tqdm, "SAMPLE_RATE": 16000, "MAX_RMS": 0.005,
          "MESSAGE_BITS": 16}
    exec(compile(ast.Module(body=nodes, type_ignores=[]), "<reference main16 extract>", "exec"), ns)
    return ns


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    args = ap.parse_args()
    torch.set_num_threads(8)
    ref = extract(args.ref)
    gsd, dsd = R.reference_layout_init()
    R.perturb_bn_(gsd, seed=R.BN_SEED_G)
    R.perturb_bn_(dsd, seed=R.BN_SEED_D)
    refG, refD = ref["Generator"](message_bits=16), ref["Detector"](message_bits=16)
    refG.load_state_dict(gsd); refD.load_state_dict(dsd)
    fx, report = OrderedDict(), []

    # ---- evaluate_model on two batches (4 + 2 clips: a ragged last batch), messages drawn by the reference's own randint
    batches = [O.synthetic_clips(4, seed=501), O.synthetic_clips(2, seed=502)]
    torch.manual_seed(EVAL_SEED)
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        res = ref["evaluate_model"](refG, refD, batches, torch.device("cpu"))
    torch.manual_seed(EVAL_SEED)
    msgs = [torch.randint(0, 2 ** 16, (b.shape[0],)) for b in batches]        # the same draws, in the same order
    per = [O.evaluate_batch(gsd, dsd, b, m) for b, m in zip(batches, msgs)]
    mine = {"watermarked_prob": np.mean(np.concatenate([p["prob_watermarked"].numpy() for p in per])),
            "clean_prob": np.mean(np.concatenate([p["prob_clean"].numpy() for p in per])),
            "bit_accuracy": np.mean(np.concatenate([p["bit_accuracy"].numpy() for p in per])),
            "delta_rms": np.mean(np.concatenate([p["delta_rms"].numpy() for p in per]))}
    for k in res:
        report.append((f"evaluate_model {k}", abs(float(res[k]) - float(mine[k]))))
        fx[f"eval_{k}"] = np.array(float(res[k]), dtype=np.float64)
    fx["eval_seed"] = np.array(EVAL_SEED)
    fx["eval_messages"] = torch.cat(msgs).numpy()
    fx["eval_clip_prob_watermarked"] = np.concatenate([p["prob_watermarked"].numpy() for p in per])
    fx["eval_clip_prob_clean"] = np.concatenate([p["prob_clean"].numpy() for p in per])
    fx["eval_clip_bit_accuracy"] = np.concatenate([p["bit_accuracy"].numpy() for p in per])
    fx["eval_clip_delta_rms"] = np.concatenate([p["delta_rms"].numpy() for p in per])

    # ---- compute_si_snr: the (1,N) waveform call and the (1,1,T) segment call of evaluate_unseen_file
    a = O.synthetic_clips(1, seed=503).reshape(1, -1)
    b = a + 0.01 * torch.randn(1, 16000, generator=torch.Generator().manual_seed(504))
    for nm, (x, y) in (("2d", (a, b)), ("3d", (a.unsqueeze(0), b.unsqueeze(0)))):
        r, o = ref["compute_si_snr"](x, y), O.compute_si_snr(x, y)
        report.append((f"compute_si_snr {nm} (ref {r})", 0.0 if (r == o) else abs(r - o)))
        fx[f"si_snr_{nm}"] = np.array(r, dtype=np.float64)

    # ---- per-segment mean detection probability on a zero-padded tail segment (detect_prob / evaluate_unseen_file body)
    refD.eval()
    seg = F.pad(a[:, :5000], (0, 11000)).unsqueeze(0)
    with torch.no_grad():
        pr = torch.sigmoid(refD(seg)[:, :, 0]).mean().item()
    po = O.detect_prob_waveform(dsd, a[:, :5000])
    report.append(("padded-tail segment mean prob", abs(pr - po)))
    fx["seg_mean_prob"] = np.array(pr, dtype=np.float64)

    bad = False
    print("=== oracle vs reference (abs diff) ===")
    for k, v in report:
        flag = "" if v <= 2e-6 else "   <-- exceeds 2e-6"
        bad |= bool(flag)
        print(f"  {k:60s} {v:.3e}{flag}")
    np.savez_compressed(os.path.join(HERE, "main16_eval_golden.npz"), **fx)
    with open(os.path.join(HERE, "oracle_vs_reference_report_eval.txt"), "w") as f:
        f.write("oracle/wm_oracle.py (N1/N3 callers) vs AST-extracted /root/reference/py/main16.py, CPU fp32, abs diff\n")
        for k, v in report:
            f.write(f"{k:60s} {v:.3e}\n")
    if bad:
        sys.exit(1)


if __name__ == "__main__":
    main()
