"""CPU oracle for the main14b_2 deep-residual variant (BASELINE config 5)  --  TEST INFRASTRUCTURE ONLY.

Functional fp32 torch restatement of py/main14b_2.py:83-224 (ResidualBlock / Generator / Detector): state dict in,
tensors out.  Pinned against the reference's own (AST-extracted) classes by tests/golden/make_golden_14b2.py.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

STRIDES = (2, 4, 5, 8)          # py/main14b_2.py:47
Tensor = torch.Tensor
State = Dict[str, Tensor]


def residual_block(sd: State, p: str, x: Tensor, stride: int) -> Tensor:
    """ResidualBlock.forward, py/main14b_2.py:95-102 (ELU, no BatchNorm; 1x1 strided skip conv when the shape changes)."""
    out = F.elu(F.conv1d(x, sd[p + "conv1.weight"], sd[p + "conv1.bias"], stride=stride, padding=1))
    out = F.conv1d(out, sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=1)
    res = x
    if (p + "skip_conv.weight") in sd:
        res = F.conv1d(x, sd[p + "skip_conv.weight"], sd[p + "skip_conv.bias"], stride=stride)
    return F.elu(out + res)


def _encoder(sd: State, x: Tensor) -> Tensor:
    x = F.conv1d(x, sd["init_conv.weight"], sd["init_conv.bias"], padding=3)
    for i, st in enumerate(STRIDES):
        x = residual_block(sd, f"encoder_blocks.{i}.", x, st)
    return x


def _decoder(sd: State, prefix: str, x: Tensor) -> Tensor:
    for i, st in enumerate(reversed(STRIDES)):
        x = F.conv_transpose1d(x, sd[f"{prefix}.{2 * i}.weight"], sd[f"{prefix}.{2 * i}.bias"], stride=st, padding=st // 2)
        x = residual_block(sd, f"{prefix}.{2 * i + 1}.", x, 1)
    return x


def _fit_length(y: Tensor, T: int) -> Tensor:
    if y.shape[-1] > T:
        return y[:, :, :T]
    if y.shape[-1] < T:
        return F.pad(y, (0, T - y.shape[-1]))
    return y


def generator_forward(sd: State, s: Tensor, message: Optional[Tensor] = None, taps: Optional[dict] = None) -> Tensor:
    """Generator.forward, py/main14b_2.py:150-177.  The message embedding is added BEFORE the LSTM (:159-163)."""
    T = s.shape[-1]
    x = _encoder(sd, s)                                                   # (B,512,50)
    if taps is not None:
        taps["enc"] = x
    xt = F.linear(x.transpose(1, 2), sd["proj.weight"], sd["proj.bias"])   # (B,50,hd)
    if message is not None:
        xt = xt + sd["E.weight"][message].unsqueeze(1)
    hd = sd["proj.weight"].shape[0]
    B = s.shape[0]
    params = [sd[f"lstm.{n}_l{l}"] for l in range(2) for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    h0 = xt.new_zeros(2, B, hd)
    out, _, _ = torch.lstm(xt, (h0, h0.clone()), params, True, 2, 0.0, False, False, True)
    if taps is not None:
        taps["lstm"] = out
    lat = F.conv1d(out.transpose(1, 2), sd["final_conv_enc.weight"], sd["final_conv_enc.bias"], padding=3)
    x = _decoder(sd, "decoder_blocks", lat)
    if taps is not None:
        taps["dec"] = x
    delta = F.conv1d(x, sd["final_conv_dec.weight"], sd["final_conv_dec.bias"], padding=3)
    return _fit_length(delta, T)


def detector_forward(sd: State, x: Tensor) -> Tensor:
    """Detector.forward, py/main14b_2.py:212-224: channel-first raw logits (B, 1+bits, T)."""
    T = x.shape[-1]
    y = _encoder(sd, x)
    y = _decoder(sd, "upsample_blocks", y)
    y = F.conv1d(y, sd["final_conv.weight"], sd["final_conv.bias"], padding=3)
    return _fit_length(y, T)


# loss weights, py/main14b_2.py:34-38
LAMBDA_L1, LAMBDA_MSSPEC, LAMBDA_LOUD, LAMBDA_LOC, LAMBDA_DEC = 0.1, 2.0, 10.0, 10.0, 1.0


def step_losses(gsd: State, dsd: State, s: Tensor, message: Tensor):
    """The loop body of train_one_epoch, py/main14b_2.py:300-352 (clamped s_w, channel-first logits, five loss terms)."""
    from collections import OrderedDict
    from . import wm_oracle as O
    B, T = s.shape[0], s.shape[-1]
    delta = generator_forward(gsd, s, message)
    s_w = torch.clamp(s + delta, -1.0, 1.0)
    logits = detector_forward(dsd, torch.cat([s_w, s], dim=0))
    det, dec = logits[:, 0, :], logits[:B, 1:, :]
    tgt = torch.cat([torch.ones(B, T), torch.zeros(B, T)], dim=0).to(s.dtype)
    loc = F.binary_cross_entropy_with_logits(det, tgt)
    bits = O.message_bits_target(message, dec.shape[1]).to(s.dtype).unsqueeze(2).expand(-1, -1, T)
    bce = F.binary_cross_entropy_with_logits(dec, bits)
    l1 = delta.abs().mean()
    mel = O.mel_loss(s, s_w)
    loud = O.loudness_loss(s, s_w)
    raw = l1 + mel + loud + loc + bce
    total = LAMBDA_L1 * l1 + LAMBDA_MSSPEC * mel + LAMBDA_LOUD * loud + LAMBDA_LOC * loc + LAMBDA_DEC * bce
    return total, OrderedDict(delta=delta, s_w=s_w, logits=logits, l1=l1, mel=mel, loud=loud, loc=loc, bce=bce,
                              raw_total=raw, total=total)
