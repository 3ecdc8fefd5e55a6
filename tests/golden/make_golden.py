#!/usr/bin/env python
"""Generate the golden fixtures in this directory from the REFERENCE itself.

Runs only in the build container (needs /root/reference, which never travels to the
GPU box).  The reference's py/main16.py is a notebook export that cannot be imported
(module-level dataset globbing / training loop / torchaudio import), so its own
top-level ``def`` / ``class`` nodes for the hot path are AST-extracted at run time and
executed in a private namespace -- nothing of the reference's text is written to disk.

What gets committed is numbers only: seeded inputs, the reference's outputs, and the
recipe (seeds) that regenerates the weights.  ``tests/test_oracle_golden.py`` then pins
``oracle/wm_oracle.py`` to these numbers on any machine.

Usage:  python tests/golden/make_golden.py  [--ref /root/reference]
"""
from __future__ import annotations

import argparse
import ast
import hashlib
import math
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
from oracle import recipes as R    # noqa: E402

WANTED = ["fir_lowpass", "clamp_peak", "limit_rms", "high_freq_penalty", "ResBlock", "Generator", "Detector",
          "TFLoudnessLoss"]


def extract_reference(ref_root: str) -> dict:
    src = open(os.path.join(ref_root, "py", "main16.py")).read()
    tree = ast.parse(src)
    seen, nodes = set(), []
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in WANTED and node.name not in seen:
            seen.add(node.name)
            nodes.append(node)
    missing = set(WANTED) - seen
    assert not missing, f"reference lacks {missing}"
    ns = {"torch": torch, "nn": nn, "F": F, "math": math, "SAMPLE_RATE": 16000, "MAX_RMS": 0.005}
    exec(compile(ast.Module(body=nodes, type_ignores=[]), "<reference main16 extract>", "exec"), ns)
    return ns


def sd_sha(sd) -> str:
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(v.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def maxdiff(a, b) -> float:
    """max |a-b| relative to max(1, max|a|): absolute for O(1) values, relative for large ones"""
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).abs().max() / max(1.0, float(a.abs().max())))


def sub(t: torch.Tensor, step: int = 97) -> np.ndarray:
    """every `step`-th frame along the last axis"""
    return t.detach()[..., ::step].contiguous().numpy()


def checksums(t: torch.Tensor) -> np.ndarray:
    d = t.detach().double()
    return np.array([d.sum().item(), d.abs().sum().item(), d.abs().max().item()], dtype=np.float64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    args = ap.parse_args()
    torch.set_num_threads(8)
    ref = extract_reference(args.ref)
    report = []

    # ------------------------------------------------------------------ weights recipe
    torch.manual_seed(R.WEIGHT_SEED)
    refG = ref["Generator"](message_bits=16)
    refD = ref["Detector"](message_bits=16)
    init_sha_g, init_sha_d = sd_sha(refG.state_dict()), sd_sha(refD.state_dict())
    gsd, dsd = R.reference_layout_init()              # our own construction, same RNG order
    assert sd_sha(gsd) == init_sha_g and sd_sha(dsd) == init_sha_d, "init recipe differs from reference init"
    assert list(gsd.keys()) == list(refG.state_dict().keys())
    assert list(dsd.keys()) == list(refD.state_dict().keys())
    R.perturb_bn_(gsd, seed=R.BN_SEED_G)
    R.perturb_bn_(dsd, seed=R.BN_SEED_D)
    refG.load_state_dict(gsd)
    refD.load_state_dict(dsd)

    fx = OrderedDict()
    fx["init_sha_g"] = np.frombuffer(bytes.fromhex(init_sha_g), dtype=np.uint8)
    fx["init_sha_d"] = np.frombuffer(bytes.fromhex(init_sha_d), dtype=np.uint8)
    fx["state_sha_g"] = np.frombuffer(bytes.fromhex(sd_sha(gsd)), dtype=np.uint8)
    fx["state_sha_d"] = np.frombuffer(bytes.fromhex(sd_sha(dsd)), dtype=np.uint8)

    # ------------------------------------------------------------------ G1: eval-mode forward, B=2
    s = O.synthetic_clips(2, seed=1234)
    msg = torch.tensor([5, 40000], dtype=torch.int64)
    refG.eval(); refD.eval()
    with torch.no_grad():
        d_ref = refG(s, msg)
        d_nomsg_ref = refG(s)                                   # message omitted, py/main16.py:437
        dp_ref = ref["limit_rms"](ref["clamp_peak"](ref["fir_lowpass"](d_ref)))
        lg_ref = refD(torch.cat([s + dp_ref, s], 0))
        taps = {}
        d_or = O.generator_forward(gsd, s, msg, training=False, taps=taps)
        d_nomsg_or = O.generator_forward(gsd, s, None, training=False)
        dp_or = O.postprocess(d_or)
        lg_or = O.detector_forward(dsd, torch.cat([s + dp_or, s], 0), training=False)
        # explicit-loop LSTM cross-check on a short prefix (independent of aten::lstm)
        xs = taps["enc"][:, :, :400].permute(0, 2, 1)
        h_loop = O.lstm_forward(xs, gsd["lstm.weight_ih_l0"], gsd["lstm.weight_hh_l0"], gsd["lstm.bias_ih_l0"],
                                gsd["lstm.bias_hh_l0"]).permute(0, 2, 1)
    report += [("G1 delta", maxdiff(d_ref, d_or)), ("G1 delta(no msg)", maxdiff(d_nomsg_ref, d_nomsg_or)),
               ("G1 delta_post", maxdiff(dp_ref, dp_or)), ("G1 logits", maxdiff(lg_ref, lg_or)),
               ("G1 lstm loop vs aten (400 steps)", maxdiff(h_loop, taps["lstm"][:, :, :400]))]
    fx["g1_message"] = msg.numpy()
    fx["g1_delta"] = d_ref.numpy()
    fx["g1_delta_nomsg_sub"] = sub(d_nomsg_ref)
    fx["g1_delta_post"] = dp_ref.numpy()
    fx["g1_logits_sub"] = lg_ref[:, ::97, :].contiguous().numpy()
    fx["g1_logits_chk"] = checksums(lg_ref)
    for k in ("enc0", "enc", "lstm", "dec0", "dec1"):
        fx[f"g1_tap_{k}_sub"] = sub(taps[k])
    fx["g1_tap_lstm_tail"] = taps["lstm"][:, :, -64:].contiguous().numpy()      # G5: long-horizon LSTM

    # ------------------------------------------------------------------ G2: train-mode step, B=4
    B = 4
    # clamp_peak's derivative is discontinuous at |f| = 0.02: a sample sitting within fp32 round-off of the
    # threshold flips between "gradient passes" and "gradient blocked" across implementations (seen: ONE such
    # sample moved every Generator gradient by ~1e-2).  Pick the first clip seed whose FIR output keeps a
    # margin of >= 1e-5 * max|f| (about 5x the fp32 noise of f) from the threshold, and record it in the fixture.
    for g2_seed in range(1235, 1400):
        s4 = O.synthetic_clips(B, seed=g2_seed)
        msg4 = O.synthetic_messages(B, seed=4322)
        with torch.no_grad():
            f_probe = O.fir_lowpass(O.generator_forward(gsd, s4, msg4, training=True))
        margin = float((f_probe.abs() - 0.02).abs().min())
        if margin >= 1e-5 * float(f_probe.abs().max()):
            break
    print(f"G2 clip seed {g2_seed}: clamp margin {margin:.2e}, clamped fraction {float((f_probe.abs() > 0.02).float().mean()):.3f}")
    fx["g2_seed"] = np.array(g2_seed)
    fx["g2_clamp_margin"] = np.array(margin)
    refG.train(); refD.train()
    refG.zero_grad(); refD.zero_grad()
    loud_mod = ref["TFLoudnessLoss"]()
    delta = refG(s4, msg4)
    delta_raw_ref = delta
    delta = ref["limit_rms"](ref["clamp_peak"](ref["fir_lowpass"](delta)))
    s_w = s4 + delta
    logits = refD(torch.cat([s_w, s4], dim=0))
    det, dec = logits[:, :, 0], logits[:B, :, 1:]
    tgt = torch.cat([torch.ones(B, 16000), torch.zeros(B, 16000)], 0)
    loc = F.binary_cross_entropy_with_logits(det, tgt)
    bitmask = (1 << torch.arange(16))
    tb = ((msg4.unsqueeze(1) & bitmask) > 0).float().unsqueeze(1).expand(-1, 16000, -1)
    bce = F.binary_cross_entropy_with_logits(dec, tb)
    l1 = F.l1_loss(delta, torch.zeros_like(delta))
    mel = O.mel_loss(s4, s_w)        # torchaudio absent: restated mel (parity unpinned), see oracle header
    loud = loud_mod(s4, s_w)
    hf = ref["high_freq_penalty"](delta)
    total = 1.0 * l1 + 4.0 * mel + 20.0 * loud + 10.0 * loc + 1.0 * bce + 5.0 * hf
    total.backward()
    ref_grads = {("g." + k): v.grad.clone() for k, v in refG.named_parameters()}
    ref_grads.update({("d." + k): v.grad.clone() for k, v in refD.named_parameters()})
    ref_new_g = {k: v.clone() for k, v in refG.state_dict().items() if "running" in k or "num_batches" in k}
    ref_new_d = {k: v.clone() for k, v in refD.state_dict().items() if "running" in k or "num_batches" in k}

    g2 = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in gsd.items()}
    d2 = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in dsd.items()}
    gst, dst = {}, {}
    tot_or, out = O.step_losses(g2, d2, s4, msg4, training=True, g_stats=gst, d_stats=dst)
    tot_or.backward()
    report += [("G2 delta_raw", maxdiff(delta_raw_ref, out["delta_raw"])), ("G2 delta", maxdiff(delta, out["delta"])),
               ("G2 logits", maxdiff(logits, out["logits"]))]
    for name, refv in (("l1", l1), ("mel", mel), ("loud", loud), ("loc", loc), ("bce", bce), ("hf", hf), ("total", total)):
        report.append((f"G2 {name}", maxdiff(refv, out[name])))
        fx[f"g2_{name}"] = np.array(refv.item(), dtype=np.float64)
    worst, worst_k, dead = 0.0, "", []
    for k, v in ref_grads.items():
        mine = (g2 if k.startswith("g.") else d2)[k[2:]].grad
        if k.endswith("block.0.bias") or k.endswith("block.3.bias"):
            # a conv bias in front of a train-mode BatchNorm has an exactly-zero true gradient;
            # both sides hold only fp32 cancellation noise there -> compare against the weight-grad scale
            dead.append(float(v.abs().max()))
            continue
        rel = float((v.double() - mine.double()).abs().max()) / (float(v.abs().max()) + 1e-12)
        if rel > worst:
            worst, worst_k = rel, k
    report.append((f"G2 worst grad rel-to-max diff ({worst_k})", worst))
    report.append(("G2 max |grad| of BN-shadowed conv biases (true value 0)", max(dead)))
    for k, v in ref_new_g.items():
        report.append((f"G2 new stat g.{k}", maxdiff(v.float(), gst[k].float())))
    for k, v in ref_new_d.items():
        report.append((f"G2 new stat d.{k}", maxdiff(v.float(), dst[k].float())))
    fx["g2_message"] = msg4.numpy()
    fx["g2_delta_raw_sub"] = sub(delta_raw_ref)
    fx["g2_delta_sub"] = sub(delta)
    fx["g2_logits_sub"] = logits.detach()[:, ::97, :].contiguous().numpy()
    fx["g2_grad_g_encoder0_weight"] = ref_grads["g.encoder.0.weight"].numpy()
    fx["g2_grad_g_lstm_whh"] = ref_grads["g.lstm.weight_hh_l0"].numpy()
    fx["g2_grad_g_lstm_wih"] = ref_grads["g.lstm.weight_ih_l0"].numpy()
    fx["g2_grad_g_emb_rows"] = ref_grads["g.embedding.weight"][msg4].numpy()
    fx["g2_grad_g_dec0_weight_sub"] = ref_grads["g.decoder.0.weight"][::4, ::4].contiguous().numpy()
    fx["g2_grad_g_enc1_b0_weight_sub"] = ref_grads["g.encoder.1.block.0.weight"][::4, ::4].contiguous().numpy()
    fx["g2_grad_g_enc1_bn1_weight"] = ref_grads["g.encoder.1.block.1.weight"].numpy()
    fx["g2_grad_d_model3_weight"] = ref_grads["d.model.3.weight"].numpy()
    fx["g2_grad_d_model0_weight"] = ref_grads["d.model.0.weight"].numpy()
    fx["g2_grad_d_m1_bn4_bias"] = ref_grads["d.model.1.block.4.bias"].numpy()
    fx["g2_new_g_enc1_bn1_rm"] = ref_new_g["encoder.1.block.1.running_mean"].numpy()
    fx["g2_new_g_enc1_bn1_rv"] = ref_new_g["encoder.1.block.1.running_var"].numpy()
    fx["g2_new_d_m2_bn4_rm"] = ref_new_d["model.2.block.4.running_mean"].numpy()
    fx["g2_new_d_m2_bn4_rv"] = ref_new_d["model.2.block.4.running_var"].numpy()

    # ------------------------------------------------------------------ G3: shipped Detector checkpoint, eval
    ck = torch.load(os.path.join(args.ref, "models", "detector_best.pth"), map_location="cpu", weights_only=True)
    ck = OrderedDict((k[len("_orig_mod."):] if k.startswith("_orig_mod.") else k, v) for k, v in ck.items())
    refD2 = ref["Detector"](message_bits=16)
    missing = refD2.load_state_dict(ck, strict=True)
    refD2.eval()
    with torch.no_grad():
        lg3_ref = refD2(s)
        lg3_or = O.detector_forward(ck, s, training=False)
    report.append(("G3 logits (shipped detector ckpt)", maxdiff(lg3_ref, lg3_or)))
    np.savez_compressed(os.path.join(HERE, "detector_best_unprefixed.npz"), **{k: v.numpy() for k, v in ck.items()})
    fx["g3_logits_sub"] = lg3_ref[:, ::97, :].contiguous().numpy()
    fx["g3_logits_chk"] = checksums(lg3_ref)
    fx["g3_mean_prob"] = torch.sigmoid(lg3_ref[:, :, 0]).mean(dim=1).numpy()

    # ------------------------------------------------------------------ G4: post-processing + loss stack alone
    with torch.no_grad():
        dbig = 0.03 * torch.randn(3, 1, 16000, generator=torch.Generator().manual_seed(77))   # exercises clamp + rms cap
        dsmall = 0.001 * torch.randn(3, 1, 16000, generator=torch.Generator().manual_seed(78))  # gain == 1 branch
        for nm, d in (("big", dbig), ("small", dsmall)):
            fr, fo = ref["fir_lowpass"](d), O.fir_lowpass(d)
            pr = ref["limit_rms"](ref["clamp_peak"](fr))
            po = O.postprocess(d)
            report += [(f"G4 fir {nm}", maxdiff(fr, fo)), (f"G4 post {nm}", maxdiff(pr, po)),
                       (f"G4 hf {nm}", maxdiff(ref["high_freq_penalty"](pr), O.high_freq_penalty(po)))]
            fx[f"g4_post_{nm}_sub"] = sub(pr)
            fx[f"g4_hf_{nm}"] = np.array(ref["high_freq_penalty"](pr).item())
        s3 = O.synthetic_clips(3, seed=99)
        lr_, lo_ = loud_mod(s3, s3 + dbig.clamp(-0.02, 0.02)), O.loudness_loss(s3, s3 + dbig.clamp(-0.02, 0.02))
        report.append(("G4 loud", maxdiff(lr_, lo_)))
        fx["g4_loud"] = np.array(lr_.item())
        fx["g4_mel_restated"] = np.array(O.mel_loss(s3, s3 + dbig.clamp(-0.02, 0.02)).item())
        fx["g4_fir_kernel"] = O.fir_kernel().numpy()
    # mel filterbank cross-check (independent implementation available offline)
    try:
        from transformers.audio_utils import mel_filter_bank
        fb2 = mel_filter_bank(513, 64, 0.0, 8000.0, 16000, norm=None, mel_scale="htk")
        report.append(("mel fbank vs transformers.audio_utils", float(np.abs(O.mel_filterbank().numpy() - fb2).max())))
    except Exception as e:  # pragma: no cover
        report.append((f"mel fbank cross-check skipped: {e}", float("nan")))
    fx["mel_fbank_colsum"] = O.mel_filterbank().sum(0).numpy()

    print("\n=== oracle vs reference (max abs diff) ===")
    bad = False
    for k, v in report:
        flag = ""
        if not (v <= 2e-6) and "transformers" not in k and "grad" not in k and not k.startswith("mel fbank"):
            flag = "   <-- exceeds 2e-6"; bad = True
        if "grad" in k and "true value 0" not in k and not (v <= 1e-4):
            flag = "   <-- exceeds 1e-4"; bad = True
        print(f"  {k:45s} {v:.3e}{flag}")
    out_path = os.path.join(HERE, "main16_golden.npz")
    np.savez_compressed(out_path, **fx)
    print("wrote", out_path, f"{os.path.getsize(out_path)/1024:.1f} KiB")
    with open(os.path.join(HERE, "oracle_vs_reference_report.txt"), "w") as f:
        f.write("oracle/wm_oracle.py vs AST-extracted /root/reference/py/main16.py, CPU fp32, max abs diff\n")
        for k, v in report:
            f.write(f"{k:45s} {v:.3e}\n")
    if bad:
        sys.exit(1)


if __name__ == "__main__":
    main()
