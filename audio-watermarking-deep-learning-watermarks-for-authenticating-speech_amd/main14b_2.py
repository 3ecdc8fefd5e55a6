"""Drop-in mirrors of the reference's deep-residual variant, py/main14b_2.py:83-224 (BASELINE config 5):
ResidualBlock / Generator / Detector with the reference's constructor arguments, sub-module names and state_dict
layout; the arithmetic runs in csrc/gconv.hip.

Status (round 1): FORWARD ONLY (inference / evaluation).  The backward kernels of this variant are not built yet, so
the modules refuse to run with autograd recording instead of silently producing a graph-less tensor.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from ._lib import lib
from .ops import _f32, _p, _stream

CHANNELS, HIDDEN_DIM, NUM_BITS, OUTPUT_CH = 32, 32, 16, 128     # py/main14b_2.py:43-46
STRIDES = [2, 4, 5, 8]                                          # :47


def _no_grad_only(mod):
    if torch.is_grad_enabled() and any(p.requires_grad for p in mod.parameters()):
        raise NotImplementedError("main14b_2 variant: only the forward path is built (run under torch.no_grad()); "
                                  "the backward kernels are planned for the next round")


def _gconv(x, w, bias=None, stride=1, padding=0, act=0, res=None, vec=None):
    """Conv1d(w [Cout,Cin,K]) on channel-first frames through wm_gconv"""
    x = ops._chk(x, "input", 3)
    NB, Cin, Lin = x.shape
    Cout, _, K = w.shape
    Lout = (Lin + 2 * padding - K) // stride + 1
    wp = w.permute(1, 2, 0).reshape(Cin * K, Cout).contiguous()          # pure data movement
    y = _f32(NB, Cout, Lout, device=x.device)
    lib.wm_gconv(_p(x), _p(wp), _p(bias), _p(vec), _p(res), _p(y), NB, Cin, Lin, K, stride, padding, Cout, Lout, 1, 0, Cout,
                 Lout, act, _stream())
    return y


def _gconvT(x, w, bias, st):
    """ConvTranspose1d(w [Cin,Cout,2*st], stride st, padding st//2) = 2-tap conv onto Cout*st phase rows + pixel shuffle"""
    x = ops._chk(x, "input", 3)
    NB, Cin, Lin = x.shape
    Cout = w.shape[1]
    pad = st // 2
    Lout = (Lin - 1) * st - 2 * pad + 2 * st
    # wp[ci*2 + kk][co*st + phase]:  kk = 1 <-> tap `phase` (x[n]),  kk = 0 <-> tap `phase + st` (x[n-1])
    wp = w.reshape(Cin, Cout, 2, st).flip(2).permute(0, 2, 1, 3).reshape(Cin * 2, Cout * st).contiguous()
    y = _f32(NB, Cout, Lout, device=x.device)
    lib.wm_gconv(_p(x), _p(wp), _p(bias), None, None, _p(y), NB, Cin, Lin, 2, 1, 1, Cout * st, Lin + 1, st, pad, Cout, Lout, 0,
                 _stream())
    return y


def make_conv1d(in_ch, out_ch, kernel_size=3, stride=1, padding=1):
    return nn.Conv1d(in_ch, out_ch, kernel_size, stride=stride, padding=padding)


class ResidualBlock(nn.Module):
    """py/main14b_2.py:86-102"""

    def __init__(self, in_ch, out_ch, stride=1):
        super().__init__()
        self.downsample = (stride != 1 or in_ch != out_ch)
        self.stride = stride
        self.conv1 = make_conv1d(in_ch, out_ch, kernel_size=3, stride=stride, padding=1)
        self.conv2 = make_conv1d(out_ch, out_ch, kernel_size=3, stride=1, padding=1)
        self.elu = nn.ELU()
        if self.downsample:
            self.skip_conv = make_conv1d(in_ch, out_ch, kernel_size=1, stride=stride, padding=0)

    def forward(self, x):
        _no_grad_only(self)
        out = _gconv(x, self.conv1.weight, self.conv1.bias, self.stride, 1, act=1)
        res = _gconv(x, self.skip_conv.weight, self.skip_conv.bias, self.stride, 0) if self.downsample else x
        return _gconv(out, self.conv2.weight, self.conv2.bias, 1, 1, act=1, res=res)


def _lstm_layers(seq, lstm, H):
    """seq [T][H][B] through the layers of nn.LSTM(H, H, num_layers=L): input projection of a whole layer as one
    K=1 gconv, then one gate-GEMM + cell launch per time step"""
    T, _, B = seq.shape
    dev = seq.device
    for l in range(lstm.num_layers):
        w_ih, w_hh = getattr(lstm, f"weight_ih_l{l}"), getattr(lstm, f"weight_hh_l{l}")
        bias = (getattr(lstm, f"bias_ih_l{l}") + getattr(lstm, f"bias_hh_l{l}")).contiguous()
        xp = _gconv(seq, w_ih.unsqueeze(-1), bias)                                   # [T][4H][B]
        whhT = w_hh.t().contiguous()                                                 # [H][4H]
        hs, cs = _f32(T, H, B, device=dev), _f32(2, H, B, device=dev)
        for t in range(T):
            lib.wm_lstm_h_step_fwd(_p(xp[t]), _p(whhT), _p(hs[t - 1]) if t else None, _p(cs[(t - 1) & 1]) if t else None,
                                   _p(hs[t]), _p(cs[t & 1]), None, H, B, _stream())
        seq = hs
    return seq


class Generator(nn.Module):
    """py/main14b_2.py:104-177"""

    def __init__(self, in_channels=1, base_channels=CHANNELS, hidden_dim=HIDDEN_DIM, message_bits=NUM_BITS,
                 output_channels=OUTPUT_CH, strides=STRIDES):
        super().__init__()
        if hidden_dim % 32 != 0:
            raise ValueError("hidden_dim must be a multiple of 32 for the matrix-core gate GEMM")
        self.message_bits = message_bits
        self.hidden_dim = hidden_dim
        self.strides = list(strides)
        self.E = nn.Embedding(num_embeddings=(2 ** message_bits), embedding_dim=hidden_dim)
        self.init_conv = nn.Conv1d(in_channels, base_channels, kernel_size=7, stride=1, padding=3)
        enc_blocks, ch = [], base_channels
        for st in strides:
            enc_blocks.append(ResidualBlock(ch, ch * 2, stride=st))
            ch *= 2
        self.encoder_blocks = nn.Sequential(*enc_blocks)
        self.proj = nn.Linear(ch, hidden_dim)
        self.lstm = nn.LSTM(input_size=hidden_dim, hidden_size=hidden_dim, num_layers=2, batch_first=True, bidirectional=False)
        self.final_conv_enc = nn.Conv1d(hidden_dim, output_channels, kernel_size=7, stride=1, padding=3)
        dec_blocks, in_ch = [], output_channels
        for st in reversed(self.strides):
            out_ch = in_ch // 2
            dec_blocks.append(nn.ConvTranspose1d(in_ch, out_ch, kernel_size=2 * st, stride=st, padding=(st // 2), output_padding=0))
            dec_blocks.append(ResidualBlock(out_ch, out_ch, stride=1))
            in_ch = out_ch
        self.decoder_blocks = nn.Sequential(*dec_blocks)
        self.final_conv_dec = nn.Conv1d(in_ch, 1, kernel_size=7, stride=1, padding=3)

    def forward(self, s, message=None):
        _no_grad_only(self)
        s = ops._chk(s, "clip batch", 3)
        B, _, T = s.shape
        x = _gconv(s, self.init_conv.weight, self.init_conv.bias, 1, 3)
        x = self.encoder_blocks(x)                                                   # (B,512,T/320)
        vec = None
        if message is not None:
            vec = self.E.weight.index_select(0, message.to(torch.int64)).contiguous()   # row gather = data movement
        xt = _gconv(x, self.proj.weight.unsqueeze(-1), self.proj.bias, vec=vec)      # (B,hd,T')  proj + embedding add
        Tq = xt.shape[-1]
        seq = _f32(Tq, self.hidden_dim, B, device=s.device)
        lib.wm_permute_acl(_p(xt), _p(seq), B, self.hidden_dim, Tq, _stream())       # -> [T'][hd][B]
        seq = _lstm_layers(seq, self.lstm, self.hidden_dim)
        back = _f32(B, self.hidden_dim, Tq, device=s.device)
        lib.wm_permute_acl(_p(seq), _p(back), Tq, self.hidden_dim, B, _stream())     # -> [B][hd][T']
        x = _gconv(back, self.final_conv_enc.weight, self.final_conv_enc.bias, 1, 3)
        for i, st in enumerate(reversed(self.strides)):
            ct = self.decoder_blocks[2 * i]
            x = _gconvT(x, ct.weight, ct.bias, st)
            x = self.decoder_blocks[2 * i + 1](x)
        delta = _gconv(x, self.final_conv_dec.weight, self.final_conv_dec.bias, 1, 3)
        if delta.shape[-1] != T:                                                     # :172-177
            m = min(delta.shape[-1], T)
            delta = delta[:, :, :m]
            if m < T:
                delta = torch.nn.functional.pad(delta, (0, T - m))
        return delta.contiguous()


class Detector(nn.Module):
    """py/main14b_2.py:179-224; returns channel-first raw logits (B, 1+bits, T)"""

    def __init__(self, in_channels=1, base_channels=CHANNELS, hidden_dim=HIDDEN_DIM, message_bits=NUM_BITS, strides=STRIDES):
        super().__init__()
        self.message_bits = message_bits
        self.strides = list(strides)
        self.init_conv = nn.Conv1d(in_channels, base_channels, kernel_size=7, stride=1, padding=3)
        enc_blocks, ch = [], base_channels
        for st in strides:
            enc_blocks.append(ResidualBlock(ch, ch * 2, stride=st))
            ch *= 2
        self.encoder_blocks = nn.Sequential(*enc_blocks)
        dec_blocks, in_ch = [], ch
        for st in reversed(self.strides):
            out_ch = in_ch // 2
            dec_blocks.append(nn.ConvTranspose1d(in_ch, out_ch, kernel_size=2 * st, stride=st, padding=(st // 2), output_padding=0))
            dec_blocks.append(ResidualBlock(out_ch, out_ch, stride=1))
            in_ch = out_ch
        self.upsample_blocks = nn.Sequential(*dec_blocks)
        self.final_conv = nn.Conv1d(base_channels, 1 + message_bits, kernel_size=7, stride=1, padding=3)

    def forward(self, x):
        _no_grad_only(self)
        x = ops._chk(x, "clip batch", 3)
        T = x.shape[-1]
        y = _gconv(x, self.init_conv.weight, self.init_conv.bias, 1, 3)
        y = self.encoder_blocks(y)
        for i, st in enumerate(reversed(self.strides)):
            ct = self.upsample_blocks[2 * i]
            y = _gconvT(y, ct.weight, ct.bias, st)
            y = self.upsample_blocks[2 * i + 1](y)
        out = _gconv(y, self.final_conv.weight, self.final_conv.bias, 1, 3)
        if out.shape[-1] > T:
            out = out[:, :, :T]
        elif out.shape[-1] < T:
            out = torch.nn.functional.pad(out, (0, T - out.shape[-1]))
        return out.contiguous()
