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
    inference.save_audio(w, p, lowpass_hz=None)
    r = inference.load_audio(p)
    assert r.shape == w.shape and float((r - w).abs().max()) < 1.0 / 16000      # truncating quantiser: up to one LSB
    # the float container generate_watermarked_audio writes (py/main16.py:1051-1055): samples bit for bit
    pf = str(tmp_path / "a" / "f.wav")
    inference.save_audio_float(w, pf)
    assert torch.equal(inference.load_audio(pf), w)


def test_save_audio_pcm_is_the_reference_quantiser(tmp_path):
    """py/main15.py:850-867: save_audio(waveform, output_path, sample_rate) = lowpass_biquad 7 kHz -> clamp(-1, 1) -> x 32767 ->
    TRUNCATING .to(torch.int16) -> PCM_S 16.  Integer work: the bytes on disk must equal that expression exactly."""
    import wave
    g = torch.Generator().manual_seed(3)
    x = torch.cat([0.7 * torch.randn(1, 30000, generator=g),                    # |x| > 1 in places: the clamp matters
                   torch.tensor([[1.0, -1.0, 0.99999, -0.99999, 0.5 / 32767, -0.5 / 32767, 1.5 / 32767, -1.5 / 32767, 0.0]])], dim=1)
    # (a) no filter: bytes == (x.clamp(-1,1)*32767).to(int16); truncation toward zero, not rounding
    p0 = str(tmp_path / "q.wav")
    inference.save_audio(x, p0, 16000, lowpass_hz=None)
    with wave.open(p0, "rb") as w:
        assert (w.getnchannels(), w.getsampwidth(), w.getframerate()) == (1, 2, 16000)
        got = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
    want = (x.clamp(-1.0, 1.0) * 32767).to(torch.int16).numpy().reshape(-1)
    assert np.array_equal(got, want)
    assert list(want[-9:]) == [32767, -32767, 32766, -32766, 0, 0, 1, -1, 0]
    rounded = (x.clamp(-1.0, 1.0) * 32767).round().to(torch.int16).numpy().reshape(-1)
    assert not np.array_equal(got, rounded)                                     # the round-2 quantiser would differ
    # (b) the reference's positional order and default 7 kHz low-pass: bytes == quantiser(lowpass_biquad(x))
    p1 = str(tmp_path / "f.wav")
    inference.save_audio(x, p1, 16000)
    with wave.open(p1, "rb") as w:
        got1 = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
    f = inference.lowpass_biquad(x, 16000, cutoff_freq=7000)
    assert np.array_equal(got1, (f.clamp(-1.0, 1.0) * 32767).to(torch.int16).numpy().reshape(-1))
    # two channels interleave
    p2 = str(tmp_path / "s.wav")
    x2 = torch.stack([x[0, :100], -x[0, :100]])
    inference.save_audio(x2, p2, 16000, lowpass_hz=None)
    with wave.open(p2, "rb") as w:
        assert w.getnchannels() == 2
        got2 = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").reshape(-1, 2)
    assert np.array_equal(got2.T, inference.pcm16(x2).numpy())


def test_lowpass_biquad_is_the_published_section():
    """PARITY UNPINNED (torchaudio absent, no reference fixture): checked against the closed-form frequency response of the RBJ
    low-pass section the function documents -- |H| at DC = 1, -3 dB at the cutoff for Q = 0.707, falling beyond; output clamped."""
    sr, fc = 16000, 7000
    n = torch.arange(16000, dtype=torch.float64)
    w0 = 2 * np.pi * fc / sr
    alpha, cw = np.sin(w0) / (2 * 0.707), np.cos(w0)
    b = np.array([(1 - cw) / 2, 1 - cw, (1 - cw) / 2]) / (1 + alpha)
    a = np.array([1.0, -2 * cw / (1 + alpha), (1 - alpha) / (1 + alpha)])
    for f in (0.0, 1000.0, 7000.0, 7900.0):
        x = (0.5 * torch.cos(2 * np.pi * f / sr * n)).to(torch.float32).unsqueeze(0)
        y = inference.lowpass_biquad(x, sr, cutoff_freq=fc)
        z = np.exp(-1j * 2 * np.pi * f / sr)
        H = abs((b[0] + b[1] * z + b[2] * z * z) / (a[0] + a[1] * z + a[2] * z * z))
        amp = float(y[0, 8000:].abs().max()) / 0.5
        assert abs(amp - H) <= 2e-3 * max(H, 1e-3) + 1e-4, (f, amp, H)
    assert abs(abs((b.sum()) / a.sum()) - 1.0) < 1e-12                      # unit DC gain
    big = torch.full((1, 64), 3.0)
    assert float(inference.lowpass_biquad(big, sr, fc).abs().max()) <= 1.0   # lfilter(clamp=True)


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
