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
        with wave.open(file_path, "rb") as w:
            if w.getsampwidth() != 2:
                raise ValueError("without torchaudio only 16-bit PCM wav files can be read")
            if w.getframerate() != sample_rate:
                raise ValueError(f"without torchaudio no resampling is available (file is {w.getframerate()} Hz)")
            data = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").astype(np.float32) / 32768.0
            data = data.reshape(-1, w.getnchannels()).mean(axis=1)
        return torch.from_numpy(data.copy()).unsqueeze(0)


def save_audio(file_path, waveform, sample_rate=SAMPLE_RATE):
    """16-bit PCM wav (what torchaudio.save writes for the reference's float input by default is float; the int16 path
    of py/main15.py:850-867 is the portable one and needs no torchaudio)."""
    out_dir = os.path.dirname(file_path)
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
    pcm = (waveform.detach().cpu().clamp(-1, 1) * 32767.0).round().to(torch.int16).numpy().reshape(-1)
    with wave.open(file_path, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(sample_rate)
        w.writeframes(pcm.astype("<i2").tobytes())


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


def compute_si_snr(s, s_hat, eps=1e-8):
    """py/main16.py:764-773"""
    s = s - s.mean(dim=1, keepdim=True)
    s_hat = s_hat - s_hat.mean(dim=1, keepdim=True)
    dot = torch.sum(s * s_hat, dim=1, keepdim=True)
    alpha = dot / (torch.sum(s ** 2, dim=1, keepdim=True) + eps)
    s_target = alpha * s
    e_noise = s_hat - s_target
    return (10 * torch.log10(torch.sum(s_target ** 2, dim=1) / (torch.sum(e_noise ** 2, dim=1) + eps))).mean().item()


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
        save_audio(output_file, wm)
    return {"watermarked_waveform": wm, "delta_waveform": delta, "original_waveform": orig,
            "metrics": {"watermark_rms": watermark_rms, "si_snr_db": si_snr, "power_ratio_db": power_ratio_db}}


@torch.no_grad()
def detect_waveform(waveform, detector, detection_threshold=0.5, device="cuda", max_batch=512):
    """detect_watermark (:1114-1207) on an in-memory waveform, batched; same result dict (no plotting)."""
    detector.eval()
    segs, remainder = _segments(waveform.float())
    S = segs.shape[0]
    probs, msg_logits = [], []
    for i in range(0, S, max_batch):
        logits = detector(segs[i:i + max_batch].to(device))            # [s,T,1+bits]
        probs.append(torch.sigmoid(logits[:, :, 0]).cpu())
        if getattr(detector, "message_bits", 0) > 0:
            msg_logits.append(logits[:, :, 1:].cpu())
    probs = torch.cat(probs, dim=0)                                     # [S,T]
    n = waveform.shape[1]
    temporal = probs.reshape(-1)[:n]
    mean_prob = temporal.mean().item()
    is_wm = mean_prob > detection_threshold
    result = {"mean_probability": mean_prob, "is_watermarked": is_wm, "temporal_probs": temporal.numpy(),
              "decision": "WATERMARKED" if is_wm else "NOT WATERMARKED"}
    if msg_logits:
        ml = torch.cat(msg_logits, dim=0)                               # [S,T,bits]
        per_seg = []
        for k in range(S):                                              # per-segment mean over its valid samples (:1142,:1162)
            valid = remainder if (remainder > 0 and k == S - 1) else ml.shape[1]
            per_seg.append(ml[k, :valid].mean(dim=0))
        mean_logits = torch.stack(per_seg).mean(dim=0)
        result["predicted_message"] = (mean_logits > 0).int().tolist()
        result["message_confidence"] = torch.sigmoid(mean_logits).tolist()
    return result


def detect_watermark(input_file, detector, detection_threshold=0.5, visualize=False, device="cuda"):
    waveform = load_audio(input_file) if isinstance(input_file, (str, os.PathLike)) else input_file
    return detect_waveform(waveform, detector, detection_threshold, device)


@torch.no_grad()
def evaluate_batches(generator, detector, batches, device="cuda", message_bits=16, threshold=0.5):
    """evaluate_model (:369-423): the per-batch reductions run on the device (step.eval_forward), only four scalars
    per batch come back to the host."""
    from .step import eval_forward
    generator.eval(); detector.eval()
    acc = {"watermarked_prob": [], "clean_prob": [], "bit_accuracy": [], "delta_rms": []}
    for s in batches:
        s = s.to(device)
        message = torch.randint(0, 2 ** message_bits, (s.shape[0],), device=device)
        out = eval_forward(generator, detector, s, message)
        acc["watermarked_prob"].append(out["prob_watermarked"].mean())
        acc["clean_prob"].append(out["prob_clean"].mean())
        acc["bit_accuracy"].append(out["bit_accuracy"].mean())
        acc["delta_rms"].append(out["delta_rms"].mean())
    return {k: float(torch.stack(v).mean()) if v else math.nan for k, v in acc.items()}
