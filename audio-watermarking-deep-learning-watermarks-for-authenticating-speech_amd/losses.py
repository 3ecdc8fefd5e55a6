"""delta post-processing and the loss stack of py/main16.py:53-81, :192-217 behind the reference's names."""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import ops

SAMPLE_RATE = 16000      # py/main16.py:30
MAX_RMS = 0.005          # py/main16.py:29

_cache = {}


def _fir_taps(cutoff, taps, device):
    """The tap vector of py/main16.py:58-62, built with the same fp32 torch ops (host side, cached).
    The reference's cutoff normalisation makes this numerically an all-pass -- kept as is."""
    key = ("fir", float(cutoff), int(taps), str(device))
    if key not in _cache:
        fc = cutoff / (SAMPLE_RATE / 2)
        n = torch.arange(taps) - (taps - 1) / 2
        sinc = torch.where(n == 0, 2 * fc, torch.sin(2 * math.pi * fc * n) / (math.pi * n))
        window = 0.54 - 0.46 * torch.cos(2 * math.pi * (n + (taps - 1) / 2) / (taps - 1))
        k = sinc * window
        _cache[key] = (k / k.sum()).to(torch.float32).to(device)
    return _cache[key]


def fir_lowpass(delta, cutoff=4_000, taps=101):
    return ops.PostprocFn.apply(delta, _fir_taps(cutoff, taps, delta.device), 0.0, 0.0, 0.0, 1)


def clamp_peak(d, thr=0.02):
    return ops.PostprocFn.apply(d, _fir_taps(4000, 101, d.device), float(thr), 0.0, 0.0, 2)


def limit_rms(delta, max_rms=MAX_RMS, eps=1e-8):
    return ops.PostprocFn.apply(delta, _fir_taps(4000, 101, delta.device), 0.0, float(max_rms), float(eps), 4)


def postprocess(delta, cutoff=4_000, taps=101, thr=0.02, max_rms=MAX_RMS, eps=1e-8):
    """limit_rms(clamp_peak(fir_lowpass(delta))) in one kernel (the order of py/main16.py:245-247)."""
    return ops.PostprocFn.apply(delta, _fir_taps(cutoff, taps, delta.device), float(thr), float(max_rms), float(eps), 7)


def high_freq_penalty(delta, cutoff=3_500, n_fft=512):
    if n_fft != 512:
        raise ValueError("high_freq_penalty: the HIP path is built for the reference's n_fft=512")
    kcut = int(math.floor(cutoff * n_fft / SAMPLE_RATE)) + 1       # first bin with rfftfreq > cutoff
    return ops._SpectralLossFn.apply("hf", None, delta, kcut)


def _mel_tables(device, n_freqs=513, n_mels=64, f_min=0.0, f_max=8000.0):
    """HTK mel triangles (norm=None) as torchaudio.functional.melscale_fbanks documents them, plus the
    per-mel / per-bin support ranges the kernel iterates over."""
    key = ("mel", str(device))
    if key not in _cache:
        all_freqs = torch.linspace(0, SAMPLE_RATE // 2, n_freqs)
        m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
        m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
        m_pts = torch.linspace(m_min, m_max, n_mels + 2)
        f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
        f_diff = f_pts[1:] - f_pts[:-1]
        slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
        fb = torch.clamp(torch.min((-1.0 * slopes[:, :-2]) / f_diff[:-1], slopes[:, 2:] / f_diff[1:]), min=0.0).contiguous()
        nz = fb > 0
        klo = torch.full((n_mels,), 1, dtype=torch.int32)
        khi = torch.zeros(n_mels, dtype=torch.int32)
        for m in range(n_mels):
            idx = torch.nonzero(nz[:, m]).flatten()
            if idx.numel():
                klo[m], khi[m] = int(idx[0]), int(idx[-1])
        mlo = torch.zeros(n_freqs, dtype=torch.int32)
        for k in range(n_freqs):
            idx = torch.nonzero(nz[k]).flatten()
            if idx.numel():
                mlo[k] = int(idx[0])
                assert int(idx[-1]) - int(idx[0]) <= 2
        _cache[key] = (fb.to(device), klo.to(device), khi.to(device), mlo.to(device))
    return _cache[key]


class MultiScaleMelLoss(nn.Module):
    """py/main16.py:192-202 (single-scale log-mel L1; MelSpectrogram(16000, n_fft=1024, hop=256, n_mels=64))."""

    def forward(self, clean, watermarked):
        return ops._SpectralLossFn.apply("mel", clean, watermarked, _mel_tables(watermarked.device))


class TFLoudnessLoss(nn.Module):
    """py/main16.py:204-217."""

    def __init__(self):
        super().__init__()
        self.win_size = 2048
        self.hop = 512

    def forward(self, clean, watermarked):
        return ops._SpectralLossFn.apply("loud", clean, watermarked, None)


def detection_losses(logits, message):
    """(loc_loss, bce) of py/main16.py:252-264 for logits = detector(cat([s_w, s]))."""
    return ops.BCEFn.apply(logits, message)


def l1_to_zero(delta):
    """F.l1_loss(delta, zeros_like(delta)), py/main16.py:266."""
    return ops.L1Fn.apply(delta)
