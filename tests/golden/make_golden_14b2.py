#!/usr/bin/env python
"""Golden fixtures for the main14b_2 variant (BASELINE config 5) from the REFERENCE itself (build container only).
Same recipe as make_golden.py: AST-extract the reference's own definitions, run on CPU, store numbers + seeds."""
from __future__ import annotations

import ast
import hashlib
import os
import sys
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import wm_oracle as O  # noqa: E402
from oracle import wm_oracle_14b2 as O2  # noqa: E402

WANTED = ["make_conv1d", "ResidualBlock", "Generator", "Detector"]


def extract(ref_root):
    tree = ast.parse(open(os.path.join(ref_root, "py", "main14b_2.py")).read())
    seen, nodes = set(), []
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in WANTED and node.name not in seen:
            seen.add(node.name); nodes.append(node)
    assert seen == set(WANTED)
    ns = {"torch": torch, "nn": nn, "F": F, "CHANNELS": 32, "HIDDEN_DIM": 32, "NUM_BITS": 16, "OUTPUT_CH": 128,
          "STRIDES": [2, 4, 5, 8]}
    exec(compile(ast.Module(body=nodes, type_ignores=[]), "<reference main14b_2 extract>", "exec"), ns)
    return ns


def sha(sd):
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode()); h.update(v.detach().cpu().contiguous().numpy().tobytes())
    return np.frombuffer(h.digest(), dtype=np.uint8)


def main():
    ref = extract("/root/reference")
    fx = OrderedDict()
    report = []
    for hd in (256, 32):
        torch.manual_seed(42)
        G = ref["Generator"](hidden_dim=hd)
        D = ref["Detector"]()
        G.eval(); D.eval()
        gsd = OrderedDict((k, v.detach().clone()) for k, v in G.state_dict().items())
        dsd = OrderedDict((k, v.detach().clone()) for k, v in D.state_dict().items())
        fx[f"hd{hd}_init_sha_g"] = sha(gsd)
        fx[f"hd{hd}_init_sha_d"] = sha(dsd)
        s = O.synthetic_clips(2, seed=1234)
        msg = torch.tensor([5, 40000])
        with torch.no_grad():
            d_ref = G(s, msg)
            d0_ref = G(s)
            lg_ref = D(s + d_ref)
            taps = {}
            d_or = O2.generator_forward(gsd, s, msg, taps=taps)
            d0_or = O2.generator_forward(gsd, s, None)
            lg_or = O2.detector_forward(dsd, s + d_or)
        for nm, a, b in (("delta", d_ref, d_or), ("delta(no msg)", d0_ref, d0_or), ("logits", lg_ref, lg_or)):
            report.append((f"hd{hd} {nm}", float((a - b).abs().max() / max(1.0, float(a.abs().max())))))
        assert d_ref.shape == (2, 1, 16000) and lg_ref.shape == (2, 17, 16000)
        fx[f"hd{hd}_delta"] = d_ref.numpy()
        fx[f"hd{hd}_delta_nomsg_sub"] = d0_ref[..., ::97].contiguous().numpy()
        fx[f"hd{hd}_logits_sub"] = lg_ref[..., ::97].contiguous().numpy()
        fx[f"hd{hd}_tap_enc"] = taps["enc"][:, ::8].contiguous().numpy()
        fx[f"hd{hd}_tap_lstm"] = taps["lstm"].contiguous().numpy()
        fx[f"hd{hd}_tap_dec_sub"] = taps["dec"][..., ::97].contiguous().numpy()
        fx[f"hd{hd}_nparams"] = np.array([sum(p.numel() for p in G.parameters()), sum(p.numel() for p in D.parameters())])
    fx["message"] = np.array([5, 40000])
    print("=== oracle_14b2 vs reference (scaled max abs diff) ===")
    bad = False
    for k, v in report:
        print(f"  {k:24s} {v:.3e}")
        bad |= not (v <= 2e-6)
    out = os.path.join(HERE, "main14b2_golden.npz")
    np.savez_compressed(out, **fx)
    print("wrote", out, f"{os.path.getsize(out)/1024:.1f} KiB")
    with open(os.path.join(HERE, "oracle_vs_reference_report.txt"), "a") as f:
        f.write("\noracle/wm_oracle_14b2.py vs AST-extracted /root/reference/py/main14b_2.py\n")
        for k, v in report:
            f.write(f"{k:45s} {v:.3e}\n")
    if bad:
        sys.exit(1)


if __name__ == "__main__":
    main()
