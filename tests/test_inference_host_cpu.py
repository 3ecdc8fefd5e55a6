"""CPU-side checks of the N1 / N4 host code: segmentation, wav I/O without torchaudio, checkpoint formats."""
import os

import numpy as np
import pytest
import torch

import awm_amd
from awm_amd import checkpoint, inference


def test_segments_edge_cases():
    for n in (0, 15999, 16000, 16001, 40000):
        w = torch.arange(n, dtype=torch.float32).unsqueeze(0)
        segs, rem = inference._segments(w)
        assert rem == n % 16000
        assert segs.shape == ((n + 15999) // 16000, 1, 16000)
        if n:
            assert torch.equal(segs.reshape(1, -1)[:, :n], w)
            assert float(segs.reshape(-1)[n:].abs().sum()) == 0.0


def test_wav_roundtrip_without_torchaudio(tmp_path):
    w = (0.5 * torch.sin(torch.arange(20000) * 0.01)).unsqueeze(0)
    p = str(tmp_path / "a" / "x.wav")
    inference.save_audio(p, w)
    r = inference.load_audio(p)
    assert r.shape == w.shape and float((r - w).abs().max()) < 1.0 / 32000


def test_si_snr_formula():
    g = torch.Generator().manual_seed(0)
    s = torch.randn(1, 5000, generator=g)
    n = 0.1 * torch.randn(1, 5000, generator=g)
    v = inference.compute_si_snr(s, s + n)
    assert 18.0 < v < 22.0


def test_checkpoint_formats(tmp_path):
    G, D = awm_amd.Generator(16), awm_amd.Detector(16)
    gp, dp = str(tmp_path / "generator_best.pth"), str(tmp_path / "detector_best.pth")
    checkpoint.save_best(G, D, gp, dp, compile_prefix=True)             # as the reference's compiled models save them
    assert all(k.startswith("_orig_mod.") for k in torch.load(dp, weights_only=True))
    G2, D2 = awm_amd.Generator(16), awm_amd.Detector(16)
    assert not checkpoint.load_best(G2, gp).missing_keys and not checkpoint.load_best(D2, dp).missing_keys
    for (k, a), (_, b) in zip(D.state_dict().items(), D2.state_dict().items()):
        assert torch.equal(a, b), k
    opt = torch.optim.Adam(list(G.parameters()) + list(D.parameters()), lr=1e-3)
    rp = str(tmp_path / "ckpt_latest.pth")
    checkpoint.save_resumable(rp, epoch=3, step=1234, best_val=0.5, generator=G, detector=D, optimizer=opt)
    ck = torch.load(rp, weights_only=True)
    assert set(ck) == {"epoch", "step", "best_val", "gen", "det", "opt", "sched"}      # py/main14d.py:540-558
    G3, D3 = awm_amd.Generator(16), awm_amd.Detector(16)
    opt3 = torch.optim.Adam(list(G3.parameters()) + list(D3.parameters()), lr=1e-3)
    assert checkpoint.load_resumable(rp, G3, D3, opt3) == (3, 1234, 0.5)
    assert torch.equal(G3.state_dict()["lstm.weight_hh_l0"], G.state_dict()["lstm.weight_hh_l0"])
