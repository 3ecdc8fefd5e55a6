"""CPU oracle for the watermark embed+detect hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU, fp32) *restatement* of the algorithm of the
reference's hot path (py/main16.py Generator / Detector / delta post-processing /
loss stack).  It is the checker the HIP path is compared against.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it; the product package never does (and fails loudly without its HIP
library instead of falling back to this).

Pinning: the reference holds no tests / golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference
itself, run in the build container by ``tests/golden/make_golden.py`` (AST
extraction of the reference's own class / function definitions, executed on CPU)
and committed as numeric fixtures under ``tests/golden/``.  The one exception is
the mel-spectrogram arithmetic, which the reference delegates to torchaudio
(absent from the image): that part is restated from torchaudio's documented
defaults and is **parity unpinned** w.r.t. torchaudio itself (cross-checked
against transformers.audio_utils.mel_filter_bank only).

The style is deliberately functional: every function takes a flat state dict
(``{"encoder.0.weight": tensor, ...}``, the reference's own key names, see
SURVEY.md appendix A) instead of nn.Module objects.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

SAMPLE_RATE = 16000          # py/main16.py:30
AUDIO_LEN = 16000            # py/main16.py:31
MAX_RMS = 0.005              # py/main16.py:29
MESSAGE_BITS = 16            # py/main16.py:34
# loss weights, py/main16.py:38-43
LAMBDA_L1, LAMBDA_MSSPEC, LAMBDA_LOUD = 1.0, 4.0, 20.0
LAMBDA_LOC, LAMBDA_DEC, HF_PENALTY_W = 10.0, 1.0, 5.0
BN_EPS, BN_MOMENTUM = 1e-5, 0.1   # nn.BatchNorm1d defaults used at py/main16.py:117,120

Tensor = torch.Tensor
State = Dict[str, Tensor]


# ----------------------------------------------------------------------------
# building blocks
# ----------------------------------------------------------------------------
def _bn(sd: State, p: str, x: Tensor, training: bool, new_stats: Optional[dict]) -> Tensor:
    """BatchNorm1d over (B, T) per channel (py/main16.py:117,120).

    training=True: batch statistics (biased variance for normalisation); the
    running statistics that nn.BatchNorm1d would hold after the call (momentum
    0.1, *unbiased* variance) are returned through ``new_stats`` instead of being
    written into ``sd``.  training=False: running statistics.  Uses F.batch_norm (the
    aten kernel the reference's nn.BatchNorm1d runs) so the oracle is bit-identical to
    the reference on CPU; ``bn_explicit`` below is the same arithmetic written out.
    """
    w, b = sd[p + "weight"], sd[p + "bias"]
    rm, rv = sd[p + "running_mean"].clone(), sd[p + "running_var"].clone()
    y = F.batch_norm(x, rm, rv, w, b, training, BN_MOMENTUM, BN_EPS)
    if training and new_stats is not None:
        new_stats[p + "running_mean"] = rm
        new_stats[p + "running_var"] = rv
        new_stats[p + "num_batches_tracked"] = sd[p + "num_batches_tracked"] + 1
    return y


def bn_explicit(x: Tensor, w: Tensor, b: Tensor, rm: Tensor, rv: Tensor, training: bool):
    """The same BatchNorm written out (used by tests to cross-check F.batch_norm and by the
    fp64 'truth' runs): returns (y, new_running_mean, new_running_var)."""
    if training:
        n = x.shape[0] * x.shape[2]
        mean = x.mean(dim=(0, 2))
        var_b = x.var(dim=(0, 2), unbiased=False)
        new_rm = (1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean.detach()
        new_rv = (1 - BN_MOMENTUM) * rv + BN_MOMENTUM * var_b.detach() * (n / max(n - 1, 1))
    else:
        mean, var_b, new_rm, new_rv = rm, rv, rm, rv
    inv = torch.rsqrt(var_b + BN_EPS)
    return (x - mean[None, :, None]) * (inv * w)[None, :, None] + b[None, :, None], new_rm, new_rv


def resblock(sd: State, p: str, x: Tensor, training: bool, new_stats: Optional[dict] = None) -> Tensor:
    """ResBlock.forward, py/main16.py:112-125: relu(x + BN(conv3(relu(BN(conv3(x))))))."""
    y = F.conv1d(x, sd[p + "block.0.weight"], sd[p + "block.0.bias"], padding=1)
    y = torch.relu(_bn(sd, p + "block.1.", y, training, new_stats))
    y = F.conv1d(y, sd[p + "block.3.weight"], sd[p + "block.3.bias"], padding=1)
    y = _bn(sd, p + "block.4.", y, training, new_stats)
    return torch.relu(x + y)


def lstm_forward(x_bt: Tensor, w_ih: Tensor, w_hh: Tensor, b_ih: Tensor, b_hh: Tensor) -> Tensor:
    """Single-layer unidirectional LSTM, batch_first, zero initial state, gate order
    i,f,g,o (nn.LSTM(64,64,batch_first=True), py/main16.py:138,153).

    Written as an explicit time loop (no call into aten::lstm) so the restatement is
    independent of the library kernel it is validated against.
    """
    B, T, _ = x_bt.shape
    H = w_hh.shape[1]
    xp = x_bt @ w_ih.t() + (b_ih + b_hh)          # (B,T,4H)
    h = x_bt.new_zeros(B, H)
    c = x_bt.new_zeros(B, H)
    whh_t = w_hh.t()
    outs = []
    for t in range(T):
        a = xp[:, t] + h @ whh_t
        i, f, g, o = a.split(H, dim=1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
        h = torch.sigmoid(o) * torch.tanh(c)
        outs.append(h)
    return torch.stack(outs, dim=1)


def generator_forward(sd: State, s: Tensor, message: Optional[Tensor] = None, *, training: bool = False,
                      message_bits: int = MESSAGE_BITS, new_stats: Optional[dict] = None,
                      use_aten_lstm: bool = True, taps: Optional[dict] = None) -> Tensor:
    """Generator.forward, py/main16.py:149-162.  s (B,1,T) -> delta (B,1,T).

    ``use_aten_lstm`` selects torch's fused CPU LSTM (fast; what the reference runs)
    or the explicit loop above (slow, for cross-validation on short clips).
    ``taps`` (optional dict) receives intermediate activations for layer-wise tests.
    """
    x = F.conv1d(s, sd["encoder.0.weight"], sd["encoder.0.bias"], padding=3)      # :134
    if taps is not None:
        taps["enc0"] = x
    x = resblock(sd, "encoder.1.", x, training, new_stats)                         # :135
    x = resblock(sd, "encoder.2.", x, training, new_stats)                         # :136
    if taps is not None:
        taps["enc"] = x
    xt = x.permute(0, 2, 1)                                                        # :152
    wi, wh = sd["lstm.weight_ih_l0"], sd["lstm.weight_hh_l0"]
    bi, bh = sd["lstm.bias_ih_l0"], sd["lstm.bias_hh_l0"]
    if use_aten_lstm:
        B, H = xt.shape[0], wh.shape[1]
        h0 = xt.new_zeros(1, B, H)
        out, _, _ = torch.lstm(xt, (h0, h0.clone()), (wi, wh, bi, bh), True, 1, 0.0, False, False, True)
    else:
        out = lstm_forward(xt, wi, wh, bi, bh)
    x = out.permute(0, 2, 1)                                                       # :154
    if taps is not None:
        taps["lstm"] = x
    if message_bits > 0 and message is not None:                                   # :156-159
        x = x + sd["embedding.weight"][message].unsqueeze(-1)
    x = F.conv_transpose1d(x, sd["decoder.0.weight"], sd["decoder.0.bias"], padding=3)   # :144
    if taps is not None:
        taps["dec0"] = x
    x = resblock(sd, "decoder.1.", x, training, new_stats)                         # :145
    if taps is not None:
        taps["dec1"] = x
    return F.conv1d(x, sd["decoder.2.weight"], sd["decoder.2.bias"])               # :146


def detector_forward(sd: State, x: Tensor, *, training: bool = False, new_stats: Optional[dict] = None) -> Tensor:
    """Detector.forward, py/main16.py:183-186.  (B,1,T) -> (B,T,1+bits)."""
    y = F.conv1d(x, sd["model.0.weight"], sd["model.0.bias"], padding=3)           # :177
    y = resblock(sd, "model.1.", y, training, new_stats)                           # :178
    y = resblock(sd, "model.2.", y, training, new_stats)                           # :179
    y = F.conv1d(y, sd["model.3.weight"], sd["model.3.bias"])                      # :180
    return y.permute(0, 2, 1)                                                      # :186


# ----------------------------------------------------------------------------
# delta post-processing, py/main16.py:53-72
# ----------------------------------------------------------------------------
def fir_kernel(cutoff: float = 4000.0, taps: int = 101) -> Tensor:
    """The 101-tap 'low-pass' of py/main16.py:58-62, computed with the same fp32 torch ops.

    Quirk kept on purpose: fc = cutoff/(SR/2) = 0.5 is used as
    cycles/sample, so the sinc term is fp32 round-off off-centre and the filter is
    numerically an all-pass.
    """
    fc = cutoff / (SAMPLE_RATE / 2)
    n = torch.arange(taps) - (taps - 1) / 2
    sinc = torch.where(n == 0, 2 * fc, torch.sin(2 * math.pi * fc * n) / (math.pi * n))
    window = 0.54 - 0.46 * torch.cos(2 * math.pi * (n + (taps - 1) / 2) / (taps - 1))
    k = sinc * window
    return (k / k.sum()).to(torch.float32)


def fir_lowpass(delta: Tensor, cutoff: float = 4000.0, taps: int = 101) -> Tensor:
    k = fir_kernel(cutoff, taps).to(delta.device, delta.dtype).view(1, 1, -1)
    return F.conv1d(delta, k, padding=(taps - 1) // 2)                             # :64


def clamp_peak(d: Tensor, thr: float = 0.02) -> Tensor:
    return d.clamp(-thr, thr)                                                      # :67


def limit_rms(delta: Tensor, max_rms: float = MAX_RMS, eps: float = 1e-8) -> Tensor:
    cur = torch.sqrt((delta ** 2).mean(dim=[1, 2], keepdim=True) + eps)            # :70
    gain = torch.clamp(max_rms / cur, max=1.0)                                     # :71
    return delta * gain


def postprocess(delta: Tensor) -> Tensor:
    """fir_lowpass -> clamp_peak -> limit_rms in the order of py/main16.py:245-247."""
    return limit_rms(clamp_peak(fir_lowpass(delta)))


# ----------------------------------------------------------------------------
# loss stack
# ----------------------------------------------------------------------------
def _stft(x: Tensor, n_fft: int, hop: int) -> Tensor:
    """torch.stft defaults used by the reference: periodic hann, center=True,
    reflect pad, onesided, not normalised.  x (B,T) -> complex (B, n_fft/2+1, frames)."""
    win = torch.hann_window(n_fft, device=x.device).to(x.dtype)
    return torch.stft(x, n_fft, hop, window=win, return_complex=True)


def high_freq_penalty(delta: Tensor, cutoff: float = 3500.0, n_fft: int = 512) -> Tensor:
    """py/main16.py:74-81; masked-out bins still count in the mean's denominator."""
    spec = _stft(delta.squeeze(1), n_fft, n_fft // 4).abs()
    freqs = torch.fft.rfftfreq(n_fft, 1 / SAMPLE_RATE).to(delta.device)
    mask = (freqs > cutoff).to(spec.dtype).view(1, -1, 1)
    return (spec * mask).mean()


def mel_filterbank(n_freqs: int = 513, n_mels: int = 64, f_min: float = 0.0, f_max: float = 8000.0,
                   sample_rate: int = SAMPLE_RATE) -> Tensor:
    """HTK mel triangles, norm=None, as torchaudio.functional.melscale_fbanks documents
    them (torchaudio is what py/main16.py:195-197 calls; it is absent here => this is a
    restatement of the published algorithm, parity unpinned).  Returns (n_freqs, n_mels)."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.min(down, up), min=0.0)


def mel_spectrogram(x: Tensor, n_fft: int = 1024, hop: int = 256, n_mels: int = 64) -> Tensor:
    """MelSpectrogram(sample_rate=16000, n_fft=1024, hop_length=256, n_mels=64), power 2.
    x (B,1,T) -> (B,1,n_mels,frames)."""
    p = _stft(x.reshape(-1, x.shape[-1]), n_fft, hop).abs().pow(2.0)       # (B,F,frames)
    fb = mel_filterbank(n_fft // 2 + 1, n_mels).to(x.device, x.dtype)
    mel = torch.matmul(p.transpose(-1, -2), fb).transpose(-1, -2)
    return mel.reshape(x.shape[:-1] + mel.shape[-2:])


def mel_loss(clean: Tensor, wm: Tensor) -> Tensor:
    """MultiScaleMelLoss.forward, py/main16.py:199-202 (single scale despite the name)."""
    return F.l1_loss(torch.log(mel_spectrogram(clean) + 1e-5), torch.log(mel_spectrogram(wm) + 1e-5))


def loudness_loss(clean: Tensor, wm: Tensor) -> Tensor:
    """TFLoudnessLoss.forward, py/main16.py:210-217 (n_fft 2048, hop 512, mask from |S_clean|>0.01)."""
    sc = _stft(clean.squeeze(1), 2048, 512)
    sw = _stft(wm.squeeze(1), 2048, 512)
    mask = (sc.abs() > 0.01).to(sc.real.dtype)
    return (((sw.abs() - sc.abs()) ** 2) * mask).mean()


def message_bits_target(message: Tensor, bits: int = MESSAGE_BITS) -> Tensor:
    """bit i <-> (message & (1<<i)) != 0, py/main16.py:261-262.  (B,) int64 -> (B,bits) float."""
    bitmask = (1 << torch.arange(bits, device=message.device))
    return ((message.unsqueeze(1) & bitmask) > 0).float()


def step_losses(gsd: State, dsd: State, s: Tensor, message: Tensor, *, training: bool,
                g_stats: Optional[dict] = None, d_stats: Optional[dict] = None,
                use_aten_lstm: bool = True) -> Tuple[Tensor, "OrderedDict[str, Tensor]"]:
    """One pass of the hot loop of train_one_epoch / validate_one_epoch
    (py/main16.py:244-276 and :318-347) up to and including the weighted total.

    Returns (total_loss, dict of every intermediate the parity tests compare).
    """
    B, T = s.shape[0], s.shape[-1]
    delta_raw = generator_forward(gsd, s, message, training=training, new_stats=g_stats,
                                  use_aten_lstm=use_aten_lstm)
    delta = postprocess(delta_raw)
    s_w = s + delta
    logits = detector_forward(dsd, torch.cat([s_w, s], dim=0), training=training, new_stats=d_stats)
    det, dec = logits[:, :, 0], logits[:B, :, 1:]
    tgt = torch.cat([torch.ones(B, T), torch.zeros(B, T)], dim=0).to(s.device, s.dtype)
    loc = F.binary_cross_entropy_with_logits(det, tgt)
    bits = message_bits_target(message, dec.shape[-1]).to(s.dtype).unsqueeze(1).expand(-1, T, -1)
    bce = F.binary_cross_entropy_with_logits(dec, bits)
    l1 = delta.abs().mean()
    mel = mel_loss(s, s_w)
    loud = loudness_loss(s, s_w)
    hf = high_freq_penalty(delta)
    raw = l1 + mel + loud + loc + bce
    total = (LAMBDA_L1 * l1 + LAMBDA_MSSPEC * mel + LAMBDA_LOUD * loud +
             LAMBDA_LOC * loc + LAMBDA_DEC * bce + HF_PENALTY_W * hf)
    out = OrderedDict(delta_raw=delta_raw, delta=delta, s_w=s_w, logits=logits, l1=l1, mel=mel, loud=loud,
                      loc=loc, bce=bce, hf=hf, raw_total=raw, total=total)
    return total, out


# ----------------------------------------------------------------------------
# callers of the hot path: evaluation reductions and file-level wrappers
# ----------------------------------------------------------------------------
def evaluate_batch(gsd: State, dsd: State, s: Tensor, message: Tensor) -> "OrderedDict[str, Tensor]":
    """per-batch quantities of evaluate_model, py/main16.py:383-403 (eval mode, no grad): per-clip mean detection
    probability of the watermarked / clean halves, majority-vote bit accuracy, delta RMS."""
    B = s.shape[0]
    with torch.no_grad():
        delta = postprocess(generator_forward(gsd, s, message, training=False))
        logits = detector_forward(dsd, torch.cat([s + delta, s], dim=0), training=False)
        avg_probs = torch.sigmoid(logits[:, :, 0]).mean(dim=1)
        decoded_bits = (torch.sigmoid(logits[:B, :, 1:]) > 0.5).float().mean(dim=1) > 0.5
        acc = (decoded_bits == message_bits_target(message, logits.shape[-1] - 1)).float().mean(dim=1)
        rms = torch.sqrt((delta ** 2).mean(dim=[1, 2]))
    return OrderedDict(delta=delta, logits=logits, prob_watermarked=avg_probs[:B], prob_clean=avg_probs[B:],
                       bit_accuracy=acc, delta_rms=rms)


def compute_si_snr(s: Tensor, s_hat: Tensor, eps: float = 1e-8) -> float:
    """py/main16.py:764-773, reductions over dim=1 whatever the rank of the input.  Called on (1,N) waveforms it is the
    usual SI-SNR; evaluate_unseen_file (:1294) calls it on (1,1,T) segments, where dim=1 is the size-1 channel axis:
    s - mean == 0, so every segment yields 10*log10(0 / eps) = -inf.  Restated as is."""
    s = s - s.mean(dim=1, keepdim=True)
    s_hat = s_hat - s_hat.mean(dim=1, keepdim=True)
    dot = torch.sum(s * s_hat, dim=1, keepdim=True)
    alpha = dot / (torch.sum(s ** 2, dim=1, keepdim=True) + eps)
    s_target = alpha * s
    e_noise = s_hat - s_target
    return (10 * torch.log10(torch.sum(s_target ** 2, dim=1) / (torch.sum(e_noise ** 2, dim=1) + eps))).mean().item()


def _one_second_segments(waveform: Tensor, seg_len: int = AUDIO_LEN):
    """`for i in range(0, N, AUDIO_LEN)` with the zero-padded tail, py/main16.py:1282-1285 / :1588-1591"""
    for i in range(0, waveform.shape[1], seg_len):
        seg = waveform[:, i:i + seg_len]
        if seg.shape[1] < seg_len:
            seg = F.pad(seg, (0, seg_len - seg.shape[1]))
        yield seg.unsqueeze(0)


def evaluate_unseen_waveform(gsd: State, dsd: State, waveform: Tensor, messages: Tensor):
    """evaluate_unseen_file, py/main16.py:1263-1299, after its torchaudio load (the (1,N) mono 16 kHz waveform is the
    input here); `messages[k]` replaces the per-segment torch.randint draw of :1287.  B = 1 per segment, like the
    reference.  Returns (mean clean prob, mean watermarked prob, mean SI-SNR, mean delta RMS)."""
    clean_probs, wm_probs, si, rms = [], [], [], []
    with torch.no_grad():
        for k, seg in enumerate(_one_second_segments(waveform)):
            delta = generator_forward(gsd, seg, messages[k:k + 1], training=False)
            seg_w = seg + delta
            clean_probs.append(torch.sigmoid(detector_forward(dsd, seg, training=False)[:, :, 0]).mean().item())
            wm_probs.append(torch.sigmoid(detector_forward(dsd, seg_w, training=False)[:, :, 0]).mean().item())
            rms.append(torch.sqrt((delta ** 2).mean()).item())
            si.append(compute_si_snr(seg, seg_w))
    import numpy as np
    return np.mean(clean_probs), np.mean(wm_probs), np.mean(si), np.mean(rms)


def detect_prob_waveform(dsd: State, waveform: Tensor) -> float:
    """detect_prob, py/main16.py:1575-1596, after its torchaudio load: mean over segments of each segment's mean
    detection probability -- the zero-padded tail of the last segment counts (unlike detect_watermark, :1160-1164)."""
    probs = []
    with torch.no_grad():
        for seg in _one_second_segments(waveform):
            probs.append(torch.sigmoid(detector_forward(dsd, seg, training=False)[:, :, 0]).mean().item())
    import numpy as np
    return float(np.mean(probs))


# ----------------------------------------------------------------------------
# synthetic inputs -- shared by tests, smoke and bench
# ----------------------------------------------------------------------------
def synthetic_clips(batch: int, seed: int = 1234, T: int = AUDIO_LEN) -> Tensor:
    g = torch.Generator().manual_seed(seed)
    return (0.1 * torch.randn(batch, 1, T, generator=g)).clamp_(-0.99, 0.99)


def synthetic_messages(batch: int, seed: int = 4321, bits: int = MESSAGE_BITS) -> Tensor:
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, 2 ** bits, (batch,), generator=g, dtype=torch.int64)
