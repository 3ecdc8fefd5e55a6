"""File-level embed / detect: the reference chops a recording into 1-s segments and calls the
models at B=1 per segment in a Python loop (py/main16.py:977-1066, :1114-1207, :723-762).  Here ALL segments of a
recording go through the HIP path as one [N,1,16000] batch; per-segment random messages, remainder pad/trim and the
returned dict schemas are the reference's.  Like the reference's wrappers (and unlike its training loop) no
fir/clamp/rms post-processing is applied to delta."""
from __future__ import annotations

import math
import os
import wave

import numpy as np
import torch
import torch.nn.functional as F

SAMPLE_RATE = 16000


def load_audio(file_path, sample_rate=SAMPLE_RATE):
    """(1, N) fp32 mono waveform.  Uses torchaudio when it is installed (the reference's loader, :714-720); otherwise
    16-bit PCM .wav files at the target rate are read with the standard library."""
    try:
        import torchaudio  # noqa: F401
        waveform, sr = torchaudio.load(file_path)
        if waveform.shape[0] > 1:
            waveform = waveform.mean(dim=0, keepdim=True)
        if sr != sample_rate:
            waveform = torchaudio.transforms.Resample(sr, sample_rate)(waveform)
        return waveform
    except ImportError:
        data, rate = _read_wav(file_path)
        if rate != sample_rate:
            raise ValueError(f"without torchaudio no resampling is available (file is {rate} Hz)")
        return torch.from_numpy(data.mean(axis=1).astype(np.float32)).unsqueeze(0)


def _read_wav(path):
    """RIFF/WAVE reader for the two encodings this package writes: 16-bit signed PCM (format 1, scaled by 1/32768 as
    torchaudio.load normalises it) and 32-bit IEEE float (format 3).  Returns ((frames, channels) float32, sample rate)."""
    import struct
    with open(path, "rb") as f:
        blob = f.read()
    if blob[:4] != b"RIFF" or blob[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(blob):
        tag, size = blob[pos:pos + 4], struct.unpack("<I", blob[pos + 4:pos + 8])[0]
        body = blob[pos + 8:pos + 8 + size]
        if tag == b"fmt ":
            fmt = struct.unpack("<HHIIHH", body[:16])
        elif tag == b"data":
            data = body
        pos += 8 + size + (size & 1)
    if fmt is None or data is None:
        raise ValueError(f"{path}: missing fmt or data chunk")
    code, channels, rate, _, _, bits = fmt
    if code == 1 and bits == 16:
        x = np.frombuffer(data, dtype="<i2").astype(np.float32) / 32768.0
    elif code == 3 and bits == 32:
        x = np.frombuffer(data, dtype="<f4").astype(np.float32)
    else:
        raise ValueError("without torchaudio only 16-bit PCM and 32-bit float wav files can be read")
    return x.reshape(-1, channels), rate


def save_audio_float(waveform, output_path, sample_rate=SAMPLE_RATE):
    """py/main16.py:802-804 / :1051-1055: `torchaudio.save(path, float_waveform, sample_rate)` stores a float tensor as a
    32-bit IEEE-float WAV (samples bit for bit).  Written here as a plain RIFF file (format tag 3)."""
    import struct
    out_dir = os.path.dirname(output_path)
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
    x = waveform.detach().cpu().to(torch.float32)
    if x.dim() == 1:
        x = x.unsqueeze(0)
    ch = x.shape[0]
    payload = np.ascontiguousarray(x.numpy().T).astype("<f4").tobytes()
    hdr = (b"RIFF" + struct.pack("<I", 36 + len(payload)) + b"WAVE" + b"fmt " +
           struct.pack("<IHHIIHH", 16, 3, ch, sample_rate, sample_rate * ch * 4, ch * 4, 32) + b"data" + struct.pack("<I", len(payload)))
    with open(output_path, "wb") as f:
        f.write(hdr + payload)


def lowpass_biquad(waveform, sample_rate, cutoff_freq, Q=0.707):
    """torchaudio.functional.lowpass_biquad (imported at py/main15.py:18, called at :855), restated from its published
    definition: the RBJ cookbook low-pass section  b = ((1-cos w0)/2, 1-cos w0, (1-cos w0)/2),  a = (1+alpha, -2 cos w0, 1-alpha)
    with w0 = 2 pi cutoff / sample_rate and alpha = sin w0 / (2 Q), normalised by a0, run along the last axis as an IIR filter
    in the waveform's dtype, and the output clamped to [-1, 1] (torchaudio's `lfilter(..., clamp=True)` default).
    PARITY UNPINNED: torchaudio is absent from this image and the reference holds no fixture of this function; the order of
    the fp32 additions inside torchaudio's lfilter (direct form I) is not reproduced bit for bit -- scipy's lfilter
    (transposed direct form II) carries the recursion here."""
    from scipy.signal import lfilter
    w0 = 2.0 * math.pi * float(cutoff_freq) / float(sample_rate)
    alpha = math.sin(w0) / (2.0 * float(Q))
    cw = math.cos(w0)
    b = np.array([(1.0 - cw) / 2.0, 1.0 - cw, (1.0 - cw) / 2.0], dtype=np.float64)
    a = np.array([1.0 + alpha, -2.0 * cw, 1.0 - alpha], dtype=np.float64)
    x = waveform.detach().cpu()
    y = lfilter((b / a[0]).astype(np.float32), (a / a[0]).astype(np.float32), x.to(torch.float32).numpy(), axis=-1)
    return torch.from_numpy(np.ascontiguousarray(y, dtype=np.float32)).clamp_(-1.0, 1.0).to(x.dtype)


def pcm16(waveform):
    """float waveform -> signed 16-bit PCM exactly as py/main15.py:858: clamp to [-1, 1], scale by 32767, and the
    TRUNCATING float -> int16 conversion of `.to(torch.int16)` (toward zero; no rounding)."""
    return (waveform.detach().cpu().clamp(-1.0, 1.0) * 32767).to(torch.int16)


def save_audio(waveform, output_path, sample_rate=SAMPLE_RATE, lowpass_hz=7000):
    """py/main15.py:850-867 `save_audio(waveform, output_path, sample_rate)`: 7 kHz biquad low-pass -> clamp -> x32767 ->
    truncating int16 cast -> 16-bit signed PCM WAV.  The container is written with the standard library (`torchaudio.save(...,
    encoding="PCM_S", bits_per_sample=16)` in the reference; a mono or (C, N) waveform, samples interleaved by channel).
    `lowpass_hz=None` skips the filter (the PCM bytes are then exactly pcm16(waveform))."""
    out_dir = os.path.dirname(output_path)
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
    x = waveform.detach().cpu()
    if x.dim() == 1:
        x = x.unsqueeze(0)
    if lowpass_hz is not None:
        x = lowpass_biquad(x, sample_rate, cutoff_freq=lowpass_hz)
    pcm = pcm16(x).numpy()                                   # (C, N)
    with wave.open(output_path, "wb") as w:
        w.setnchannels(pcm.shape[0]); w.setsampwidth(2); w.setframerate(sample_rate)
        w.writeframes(np.ascontiguousarray(pcm.T).astype("<i2").tobytes())


def _segments(waveform, seg_len=SAMPLE_RATE):
    """(1,N) -> ([S,1,seg_len] batch with a zero-padded last segment, remainder length)"""
    total = waveform.shape[1]
    num_full, remainder = total // seg_len, total % seg_len
    segs = [waveform[:, i * seg_len:(i + 1) * seg_len] for i in range(num_full)]
    if remainder > 0:
        segs.append(F.pad(waveform[:, num_full * seg_len:], (0, seg_len - remainder)))
    if not segs:
        return waveform.new_zeros(0, 1, seg_len), 0
    return torch.stack(segs, dim=0), remainder


def _si_snr_db(ref, est, eps):
    """scale-invariant SNR in dB along axis 1: both signals centred, `est` split into its projection on `ref` and the rest,
    10 log10 of the energy ratio (eps added to the projection's denominator and to the residual energy, as py/main16.py:764-773)"""
    ref = ref - ref.mean(dim=1, keepdim=True)
    est = est - est.mean(dim=1, keepdim=True)
    gain = (ref * est).sum(dim=1, keepdim=True) / (ref.pow(2).sum(dim=1, keepdim=True) + eps)
    proj = gain * ref
    resid = est - proj
    return 10 * torch.log10(proj.pow(2).sum(dim=1) / (resid.pow(2).sum(dim=1) + eps))


def compute_si_snr(s, s_hat, eps=1e-8):
    """py/main16.py:764-773: mean SI-SNR (dB) of `s_hat` against `s`, as a Python float"""
    return _si_snr_db(s, s_hat, eps).mean().item()


@torch.no_grad()
def embed_waveform(waveform, generator, message_bits=16, device="cuda", messages=None, max_batch=512):
    """process_audio_file_with_delta (:723-762) on an in-memory waveform, batched.
    Returns (watermarked_waveform, delta_waveform, original_waveform), each (1, N) on the CPU."""
    generator.eval()
    segs, remainder = _segments(waveform.float())
    S = segs.shape[0]
    if S == 0:
        return waveform.clone(), torch.zeros_like(waveform), waveform
    if messages is None:       # a fresh random message per second, as :1001
        messages = torch.randint(0, 2 ** message_bits, (S,), device=device)
    deltas = []
    for i in range(0, S, max_batch):
        x = segs[i:i + max_batch].to(device)
        deltas.append(generator(x, messages[i:i + max_batch].to(device)).cpu())
    delta = torch.cat(deltas, dim=0)                    # [S,1,16000]
    wm = segs + delta
    n = waveform.shape[1]
    delta_w = delta.reshape(1, -1)[:, :n]
    wm_w = wm.reshape(1, -1)[:, :n]
    return wm_w, delta_w, waveform


def generate_watermarked_audio(input_file, generator, output_file=None, message_bits=16, device="cuda"):
    """py/main16.py:977-1066 with one batched Generator call; same result dict."""
    waveform = load_audio(input_file) if isinstance(input_file, (str, os.PathLike)) else input_file
    wm, delta, orig = embed_waveform(waveform, generator, message_bits=message_bits, device=device)
    watermark_rms = torch.sqrt((delta ** 2).mean()).item()
    si_snr = compute_si_snr(orig, wm)
    power_ratio_db = 10 * np.log10(torch.mean(orig ** 2).item() / max(torch.mean(delta ** 2).item(), 1e-30))
    if output_file:
        save_audio_float(wm, output_file)                 # :1051-1055 stores the float waveform as is
    return {"watermarked_waveform": wm, "delta_waveform": delta, "original_waveform": orig,
            "metrics": {"watermark_rms": watermark_rms, "si_snr_db": si_snr, "power_ratio_db": power_ratio_db}}


@torch.no_grad()
def detect_waveform(waveform, detector, detection_threshold=0.5, device="cuda", max_batch=512):
    """detect_watermark (:1114-1207) on an in-memory waveform, batched; same result dict (no plotting).  The
    per-segment reductions run on the device: only the temporal probability track the reference returns ((N,) floats)
    and 1+bits scalars cross to the host, never the [S,T,1+bits] logits."""
    detector.eval()
    segs, remainder = _segments(waveform.float())
    S = segs.shape[0]
    n = waveform.shape[1]
    bits = int(getattr(detector, "message_bits", 0))
    probs, seg_logit_means = [], []
    for i in range(0, S, max_batch):
        logits = detector(segs[i:i + max_batch].to(device))            # [s,T,1+bits]
        probs.append(torch.sigmoid(logits[:, :, 0]))
        if bits > 0:
            ml = logits[:, :, 1:]
            last = (i + ml.shape[0] == S) and remainder > 0
            m = ml.mean(dim=1)                                          # per-segment mean over its samples (:1142)
            if last:                                                    # the remainder segment: valid samples only (:1162)
                m = torch.cat([m[:-1], ml[-1, :remainder].mean(dim=0, keepdim=True)], dim=0)
            seg_logit_means.append(m)
    temporal = torch.cat(probs, dim=0).reshape(-1)[:n]                  # [N] on the device
    mean_prob = temporal.mean().item()
    is_wm = mean_prob > detection_threshold
    result = {"mean_probability": mean_prob, "is_watermarked": is_wm, "temporal_probs": temporal.cpu().numpy(),
              "decision": "WATERMARKED" if is_wm else "NOT WATERMARKED"}
    if seg_logit_means:
        mean_logits = torch.cat(seg_logit_means, dim=0).mean(dim=0)
        result["predicted_message"] = (mean_logits > 0).int().tolist()
        result["message_confidence"] = torch.sigmoid(mean_logits).tolist()
    return result


def detect_watermark(input_file, detector, detection_threshold=0.5, visualize=False, device="cuda"):
    waveform = load_audio(input_file) if isinstance(input_file, (str, os.PathLike)) else input_file
    return detect_waveform(waveform, detector, detection_threshold, device)


@torch.no_grad()
def detect_prob(file_path, detector, sample_rate=SAMPLE_RATE, device="cuda", max_batch=512):
    """py/main16.py:1575-1596: average over the file's 1-s segments of each segment's mean detection probability.  The
    mean of a segment runs over all 16 000 samples of the zero-PADDED tail segment too (unlike detect_watermark, which
    trims it), so this is a mean of per-segment means, not the mean of the temporal track.  One batched Detector call;
    accepts a path or an in-memory (1,N) waveform."""
    waveform = load_audio(file_path, sample_rate) if isinstance(file_path, (str, os.PathLike)) else file_path
    segs, _ = _segments(waveform.float(), sample_rate)
    if segs.shape[0] == 0:
        return float("nan")                                             # np.mean([]) in the reference
    seg_means = []
    for i in range(0, segs.shape[0], max_batch):
        logits = detector(segs[i:i + max_batch].to(device))
        seg_means.append(torch.sigmoid(logits[:, :, 0]).mean(dim=1))
    return float(torch.cat(seg_means).double().mean().item())


def _si_snr_rows(s, s_hat, eps=1e-8):
    """compute_si_snr (:764-773) applied to each (1,1,T) segment of a [S,1,T] batch, as evaluate_unseen_file does
    (:1294): the reductions run over dim=1 -- for a 3-D segment that is the size-1 CHANNEL axis, so s - mean == 0 and
    every segment yields 10*log10(0/eps) = -inf.  Kept as is (reference quirk; call compute_si_snr on (1,N) waveforms
    for a meaningful value).  Returns [S] per-segment values."""
    return _si_snr_db(s, s_hat, eps).mean(dim=1)


@torch.no_grad()
def evaluate_unseen_file(filepath, generator, detector, device="cuda", message_bits=16, messages=None, max_batch=256):
    """py/main16.py:1263-1299 with all 1-s segments of the file as one batch: returns (mean clean detection probability,
    mean watermarked detection probability, mean SI-SNR, mean delta RMS) over the segments, or four Nones when the file
    cannot be read (:1264-1267).  A fresh random message per segment (:1287) unless `messages` is given.  Accepts a path
    or an in-memory (1,N) waveform.  Per-segment reductions run on the device; four scalars come back."""
    if isinstance(filepath, (str, os.PathLike)):
        try:
            waveform = load_audio(filepath)
        except Exception:
            return None, None, None, None
    else:
        waveform = filepath
    generator.eval(); detector.eval()
    segs, _ = _segments(waveform.float())
    S = segs.shape[0]
    if S == 0:
        return (float("nan"),) * 4
    if messages is None:
        messages = torch.randint(0, 2 ** message_bits, (S,), device=device)
    clean, wm, si, rms = [], [], [], []
    for i in range(0, S, max_batch):
        seg = segs[i:i + max_batch].to(device)
        delta = generator(seg, messages[i:i + max_batch].to(device))
        seg_w = seg + delta
        p = torch.sigmoid(detector(torch.cat([seg, seg_w], dim=0))[:, :, 0]).mean(dim=1)    # eval-mode BN: rows independent
        k = seg.shape[0]
        clean.append(p[:k]); wm.append(p[k:])
        rms.append(torch.sqrt((delta ** 2).mean(dim=[1, 2])))
        si.append(_si_snr_rows(seg, seg_w))
    out = torch.stack([torch.cat(v).double().mean() for v in (clean, wm, si, rms)]).cpu()
    return tuple(float(v) for v in out)


@torch.no_grad()
def evaluate_batches(generator, detector, batches, device="cuda", message_bits=16, threshold=0.5, messages=None):
    """evaluate_model (:369-423): the per-batch reductions run on the device (step.eval_forward); the per-clip values of
    all batches are pooled and averaged once, as the reference's np.mean over its extended lists does (so a ragged last
    batch weighs by its clips).  `messages` (optional list, one tensor per batch) replaces the randint draw of :381."""
    from .step import eval_forward
    generator.eval(); detector.eval()
    keys = {"watermarked_prob": "prob_watermarked", "clean_prob": "prob_clean", "bit_accuracy": "bit_accuracy",
            "delta_rms": "delta_rms"}
    acc = {k: [] for k in keys}
    for bi, s in enumerate(batches):
        s = s.to(device)
        message = (messages[bi].to(device) if messages is not None else
                   torch.randint(0, 2 ** message_bits, (s.shape[0],), device=device))
        out = eval_forward(generator, detector, s, message)
        for k, src in keys.items():
            acc[k].append(out[src])
    return {k: float(torch.cat(v).double().mean()) if v else math.nan for k, v in acc.items()}
