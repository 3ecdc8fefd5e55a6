"""Drop-in mirrors of the reference's deep-residual variant, py/main14b_2.py:83-224 (BASELINE config 5):
ResidualBlock / Generator / Detector with the reference's constructor arguments, sub-module names and state_dict
layout.  Forward AND backward run in csrc/gconv.hip:

  * every Conv1d / Linear / ConvTranspose1d (and each of their data gradients) is one launch of the generic
    matrix-core convolution `wm_gconv` with a re-indexed weight image (re-indexing = pure data movement, done here
    with torch view ops); weight / bias gradients come from `wm_gwgrad` / `wm_channel_sum`;
  * ELU is fused into the convolution epilogue forward and is one element-wise launch backward;
  * nn.LSTM(hd, hd, num_layers=2) over the 50 latent steps: the input projection of a whole layer is one K=1 gconv, the
    recurrence is a chain of per-step launches issued by the C launcher (`wm_lstm_seq_fwd`: gate GEMM on the matrix cores +
    cell update, the kernel boundary is the step barrier), BPTT mirrors it (`wm_lstm_seq_bwd`).
"""
from __future__ import annotations

import ctypes

import torch
import torch.nn as nn

from . import ops
from ._lib import lib
from .ops import _f32, _p, _stream

CHANNELS, HIDDEN_DIM, NUM_BITS, OUTPUT_CH = 32, 32, 16, 128     # py/main14b_2.py:43-46
STRIDES = [2, 4, 5, 8]                                          # :47


# ------------------------------------------------------------------------------------------ raw launches
import os as _os0
_GCONV = {"f16x3": _os0.environ.get("WM_GCONV_F16X3", "1") == "1"}


def set_gconv_f16x3(on: bool):
    """generic convolution family (wm_gconv: forward, transposed, data gradients) for layers with Cin % 16 == 0: 1 (default) the f16
    two-piece split on the f16 matrix cores (wm_gconv_h; weights scaled from max |w|, a gradient input from max |g|), 0 native fp32
    MFMA.  WM_GCONV_F16X3=0/1 sets the default."""
    _GCONV["f16x3"] = bool(on)


def _h_ok(cin):
    """the f16 two-piece build of the generic convolution takes this layer (GEMM channel count a multiple of 16)"""
    return _GCONV["f16x3"] and cin % 16 == 0 and ops.conv_bf16x6()


def _pack_h_conv(w, mode):
    """f16 image of a Conv1d weight for wm_gconv_h without the mirror's permute / flip / contiguous launches"""
    Cout, Cin, K = w.shape
    wph = torch.empty(2 * Cout * Cin * K + 4, dtype=torch.int16, device=w.device)
    lib.wm_gconv_pack_h_conv(_p(w), _p(wph), Cout, Cin, K, mode, _stream())
    return wph


def _gconv_raw(x, wp, bias, K, S, P, Mtot, Nout, st, shp, Cout, Lout, act=0, res=None, vec=None, x2=None, nph=0, out=None, grad_in=False,
               wph=None):
    NB, Cin, Lin = x.shape
    y = _f32(NB, Cout, Lout, device=x.device) if out is None else out
    Cin_tot = Cin + (x2.shape[1] if x2 is not None else 0)
    if wph is not None or (_GCONV["f16x3"] and Cin % 16 == 0 and Cin_tot % 16 == 0 and ops.conv_bf16x6()):
        if wph is None:
            wph = torch.empty(2 * Cin_tot * K * Mtot + 4, dtype=torch.int16, device=x.device)
            lib.wm_gconv_pack_h(_p(wp), _p(wph), None, Cin_tot, K, Mtot, _stream())   # fixed scale 2^8: one launch per weight image
        gsc = ymax = None
        if grad_in:
            gsc = ops.gscale_of(x)                   # the producer's maximum when it left one (below), else one pass over x
            if x2 is not None:                       # one scale for both gradient sources: the smaller of the two
                g2 = ops.gscale_of(x2)
                gsc = torch.stack([torch.minimum(gsc[0], g2[0]), torch.maximum(gsc[1], g2[1])])
            if out is None:                          # y is a gradient too: leave max |y| for its consumer (a fresh y: every element is stored)
                ymax = torch.zeros(1, dtype=torch.float32, device=x.device)
        lib.wm_gconv_h(_p(x), _p(wph), _p(bias), _p(vec), _p(res), _p(y), NB, Cin_tot, Lin, K, S, P, Mtot, Nout, st, shp, Cout, Lout, act,
                       _p(x2), Cin if x2 is not None else 0, nph, _p(gsc), _p(ymax), _stream())
        if ymax is not None:
            ops._note_gmax(y, ymax)
        return y
    lib.wm_gconv(_p(x), _p(wp), _p(bias), _p(vec), _p(res), _p(y), NB, Cin_tot, Lin, K, S, P, Mtot, Nout, st, shp, Cout, Lout, act,
                 _p(x2), Cin if x2 is not None else 0, nph, _stream())
    return y


# ---- the 64 -> 64, k3, stride-1 convolutions of this variant (conv2 of the first encoder block, the 64-channel decoder blocks) are
# the SAME shape as main16's ResBlock convolution: they run on its bf16x6 kernels (csrc/conv64.hip, fp32-grade arithmetic on the
# bf16 matrix cores, ~1.7x the rate of the generic fp32-MFMA kernel), with ELU-family epilogues instead of BatchNorm ones.
def _c64_ok(w, stride, padding, L):
    return (tuple(w.shape) == (64, 64, 3) and stride == 1 and padding == 1 and L % 4 == 0 and L >= 4 and ops.conv_bf16x6())


def _c64_conv(x, w, mode, bias, e1, epi):
    """mode 0 forward | 1 data gradient (wm_pack_w64_bf); epi 5 elu(.+bias) | 6 elu(.+bias+e1) | 7 .*ELU'(e1) | 2 .+e1 | 3 | 0 .+bias"""
    B, _, L = x.shape
    y = torch.empty_like(x)
    lib.wm_conv64_bf(_p(x), None, _p(ops.pack_w64_bf(w, mode)), None, None, None, _p(bias), _p(e1), None, None, _p(y), None, B, L, 0, epi,
                     0, _stream())
    return y


def _conv_fwd(x, w, bias, stride, padding, act=0, res=None, vec=None):
    Cout, Cin, K = w.shape
    if vec is None and bias is not None and _c64_ok(w, stride, padding, x.shape[2]) and (act == 1 or res is None):
        if act == 1:
            return _c64_conv(x, w, 0, bias, res, 6 if res is not None else 5)
        return _c64_conv(x, w, 0, bias, None, 0)
    Lout = (x.shape[2] + 2 * padding - K) // stride + 1
    if _h_ok(Cin) and w.is_contiguous():
        return _gconv_raw(x, None, bias, K, stride, padding, Cout, Lout, 1, 0, Cout, Lout, act, res, vec, wph=_pack_h_conv(w, 0))
    wp = w.permute(1, 2, 0).reshape(Cin * K, Cout).contiguous()
    return _gconv_raw(x, wp, bias, K, stride, padding, Cout, Lout, 1, 0, Cout, Lout, act, res, vec)


def _conv_dgrad(g, w, stride, padding, Lin, res=None, act=0, out=None):
    """dL/dx of Conv1d(w [Cout,Cin,K], stride, padding) for dL/dy = g [NB,Cout,Lout]; epilogue: + res (act 0) or, act 2,
    multiplied by ELU'(res) (res = the ELU output the gradient flows into: the result is dL/dz)"""
    Cout, Cin, K = w.shape
    if out is None and _c64_ok(w, stride, padding, Lin) and g.shape[2] == Lin:
        return _c64_conv(g, w, 1, None, res, 7 if act == 2 else (2 if res is not None else 3))
    if stride == 1 and _h_ok(Cout) and w.is_contiguous():
        return _gconv_raw(g, None, None, K, 1, K - 1 - padding, Cin, Lin, 1, 0, Cin, Lin, act, res, None, None, 0, out, grad_in=True,
                          wph=_pack_h_conv(w, 1))
    if stride == 1:
        wp = w.flip(2).permute(0, 2, 1).reshape(Cout * K, Cin).contiguous()             # rows (co, kk): W[co][ci][K-1-kk]
        return _gconv_raw(g, wp, None, K, 1, K - 1 - padding, Cin, Lin, 1, 0, Cin, Lin, act, res, None, None, 0, out, grad_in=True)
    # strided conv: its data gradient is a transposed conv = Kg-tap conv onto Cin*stride phase rows + pixel shuffle
    Kg = (K + stride - 1) // stride
    wz = torch.zeros(Cout, Cin, Kg * stride, dtype=w.dtype, device=w.device)
    wz[:, :, :K] = w                                                                        # tap k = phase + m*stride
    # wp[(co*Kg + kk)][ci*stride + phase] = W[co][ci][phase + (Kg-1-kk)*stride]
    wp = wz.reshape(Cout, Cin, Kg, stride).flip(2).permute(0, 2, 1, 3).reshape(Cout * Kg, Cin * stride).contiguous()
    Nout = (Lin - 1 + padding) // stride + 1
    return _gconv_raw(g, wp, None, Kg, 1, Kg - 1, Cin * stride, Nout, stride, padding, Cin, Lin, act, res, grad_in=True)


def _strided_block_dgrad(gz1, w1, gz2, ws, stride, Lin):
    """dL/dx of a down-sampling ResidualBlock's two paths in ONE launch: conv1 (k3, stride S >= 3, padding 1) and the 1x1
    skip conv (stride S) both scatter into x -- dx[ci][t*S + k - 1] = sum_co W1[co][ci][k] gz1[co][t] (+ for k = 1:
    sum_co Ws[co][ci] gz2[co][t]), every other position is zero.  As a GEMM: rows (ci, k) = 3 phases per channel instead of
    the S phases of a transposed convolution, contraction over the 2*Cout channels of [gz1; gz2] (second source of wm_gconv)."""
    Cout, Cin, K = w1.shape
    wcat = torch.zeros(2 * Cout, Cin, K, dtype=w1.dtype, device=w1.device)      # rows: gz1's channels, then gz2's
    wcat[:Cout] = w1
    wcat[Cout:, :, 1] = ws[:, :, 0]                                             # x[t*S] is tap 1 of the padded k3 window
    wp = wcat.reshape(2 * Cout, Cin * K).contiguous()                           # [(src, co)][ci*K + k]
    NB, Lo = gz1.shape[0], gz1.shape[2]
    dx = torch.zeros(NB, Cin, Lin, dtype=torch.float32, device=gz1.device)      # positions no tap reaches stay zero
    return _gconv_raw(gz1, wp, None, 1, 1, 0, Cin * K, Lo, stride, 1, Cin, Lin, 0, None, None, gz2, K, dx, grad_in=True)


import os as _os
_FUSED_STRIDED_DGRAD = _os.environ.get("WM_FUSED_STRIDED_DGRAD", "1") == "1"      # A/B knob
_PLAN = {}


def _gwgrad_workspace(NB, Ca, Cb, La, K, device):
    """split-K partial-tile workspace of wm_gwgrad for this problem shape (size from the library's own plan)"""
    key = (NB, Ca, Cb, La, K)
    if key not in _PLAN:
        out = (ctypes.c_longlong * 1)()
        lib.wm_gwgrad_plan(NB, Ca, Cb, La, K, ctypes.addressof(out), None)
        _PLAN[key] = int(out[0])
    return _f32(_PLAN[key], device=device)


def _gwgrad_raw(A, Bx, Cb, Lb, b_clip_stride, shape, K, P, want_bias, remap=0, r1=0, r2=0):
    """G[a][b][k] = sum_{nb,t} A[nb][a][t] * Bx[nb][b][t + k - P] (+ dbias[a] = sum A): deterministic split-K GEMM.
    Bx is a raw view (Cb channels of Lb floats per clip, clips b_clip_stride floats apart)."""
    NB, Ca, La = A.shape
    G = torch.empty(shape, dtype=torch.float32, device=A.device)
    db = _f32(Ca, device=A.device) if want_bias else None
    slab = _gwgrad_workspace(NB, Ca, Cb, La, K, A.device)
    lib.wm_gwgrad(_p(A), _p(Bx), _p(G), _p(db), _p(slab), NB, Ca, Cb, La, Lb, K, P, b_clip_stride, remap, r1, r2, 0, _stream())
    return G, db


def _gather_taps(x, K, S, P, Lout, order):
    """[NB][C][Lin] -> [NB][K*C | C*K][Lout]: tap planes (order 0) / stride phases (order 1) as channels"""
    NB, C, Lin = x.shape
    y = _f32(NB, C * K, Lout, device=x.device)
    lib.wm_gather_taps(_p(x), _p(y), NB, C, Lin, K, S, P, Lout, order, _stream())
    return y


def _conv_wgrad(gz, x, w_shape, stride, padding, want_bias, planes=None):
    """dW (and db) of Conv1d(w [Cout,Cin,K], stride, padding) for dL/dy = gz and input x.  A strided convolution first lays
    the K tap planes of x out as channels (x'[k*Cin + ci][t] = x[ci][t*stride + k - padding]): its weight gradient is then a
    K = 1 stride-1 GEMM over K*Cin channels (`planes` = an already gathered x', shared with the block's 1x1 skip conv)."""
    Cout, Cin, K = w_shape
    if tuple(w_shape) == (64, 64, 3) and stride == 1 and padding == 1 and x.shape[2] % 4 == 0 and ops.conv_bf16x6():
        B, _, L = x.shape
        dw, db = torch.empty(w_shape, dtype=torch.float32, device=x.device), _f32(64, device=x.device)
        part = _f32(2 * ops.NCU * (3 * 4096 + 64), device=x.device)
        lib.wm_wgrad64_bf(_p(gz), None, None, None, None, _p(x), None, None, _p(part), _p(dw), _p(db), B, L, 0, 0, 0, _stream())
        return dw, (db if want_bias else None)
    if stride == 1:
        return _gwgrad_raw(gz, x, Cin, x.shape[2], 0, w_shape, K, padding, want_bias)
    Lout = gz.shape[2]
    if planes is None:
        planes = _gather_taps(x, K, stride, padding, Lout, 0)
    return _gwgrad_raw(gz, planes, K * Cin, Lout, 0, w_shape, 1, 0, want_bias, 1, Cin, K)


def _elu_bwd(g, y):
    dz = torch.empty_like(g)
    lib.wm_elu_bwd(_p(g), _p(y), _p(dz), g.numel(), _stream())
    return dz


# ------------------------------------------------------------------------------------------ autograd Functions
class ConvFn(torch.autograd.Function):
    """Conv1d (+ per-(clip,channel) vector + residual) (+ ELU)  --  py/main14b_2.py:83-102, :121, :134, :139, :153"""

    @staticmethod
    def forward(ctx, x, w, b, stride, padding, act, res, vec, grad_rows=None):
        x = ops._chk(x, "input", 3)
        res = res.contiguous() if res is not None else None
        y = _conv_fwd(x, w, b, stride, padding, act, res, vec)
        ctx.cfg = (stride, padding, act, res is not None, vec is not None)
        ctx.grad_rows = x.shape[0] if grad_rows is None else max(0, min(int(grad_rows), x.shape[0]))
        ctx.save_for_backward(x, w, y if act else x.new_empty(0))
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        stride, padding, act, has_res, has_vec = ctx.cfg
        gy = gy.contiguous()
        gz = _elu_bwd(gy, y) if act else gy
        dx = None
        if ctx.needs_input_grad[0]:
            if ctx.grad_rows == x.shape[0] or stride != 1:
                dx = _conv_dgrad(gz, w, stride, padding, x.shape[2])
            else:       # only clips [0, grad_rows) need an input gradient (the clean half of [watermarked; clean] is data)
                dx = torch.zeros_like(x)
                if ctx.grad_rows > 0:
                    _conv_dgrad(gz[:ctx.grad_rows], w, 1, padding, x.shape[2], out=dx[:ctx.grad_rows])
        dw, db = _conv_wgrad(gz, x, w.shape, stride, padding, True)
        dvec = None
        if has_vec:
            dvec = _f32(gz.shape[0], gz.shape[1], device=gz.device)
            lib.wm_rowsum_any(_p(gz), _p(dvec), gz.shape[0] * gz.shape[1], gz.shape[2], _stream())
        return dx, dw, db, None, None, None, (gz if has_res else None), dvec, None


class ConvTFn(torch.autograd.Function):
    """ConvTranspose1d(k = 2*st, stride st, padding st//2)  --  py/main14b_2.py:147, :202"""

    @staticmethod
    def forward(ctx, x, w, b, st):
        x = ops._chk(x, "input", 3)
        NB, Cin, Lin = x.shape
        Cout, pad = w.shape[1], st // 2
        Lout = (Lin - 1) * st - 2 * pad + 2 * st
        # wp[ci*2 + kk][co*st + phase]:  kk = 1 <-> tap `phase` (x[n]),  kk = 0 <-> tap `phase + st` (x[n-1])
        wp = w.reshape(Cin, Cout, 2, st).flip(2).permute(0, 2, 1, 3).reshape(Cin * 2, Cout * st).contiguous()
        y = _gconv_raw(x, wp, b, 2, 1, 1, Cout * st, Lin + 1, st, pad, Cout, Lout)
        ctx.st = st
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        st, pad = ctx.st, ctx.st // 2
        g = g.contiguous()
        Cin, Cout, K = w.shape
        # dx[ci][j] = sum_{co,k} g[co][j*st - pad + k] W[ci][co][k]: a strided conv over g
        wp = w.permute(1, 2, 0).reshape(Cout * K, Cin).contiguous()
        dx = _gconv_raw(g, wp, None, K, st, pad, Cin, x.shape[2], 1, 0, Cin, x.shape[2], grad_in=True)
        # dW[ci][co][q*st + ph] = sum_t x[ci][t] * g'[co*st + ph][t + q] with the stride phases of g as channels:
        # a K = 2 stride-1 GEMM over Cout*st channels
        gp = _gather_taps(g, st, st, pad, x.shape[2] + 1, 1)
        dw, _ = _gwgrad_raw(x, gp, Cout * st, x.shape[2] + 1, 0, w.shape, 2, 0, False, 2, st, 0)
        db = _f32(Cout, device=g.device)
        lib.wm_channel_sum(_p(g), _p(db), _p(_f32(64 * Cout, device=g.device)), g.shape[0], Cout, g.shape[2], 0, _stream())
        return dx, dw, db, None


class PermuteFn(torch.autograd.Function):
    """[A][C][L] -> [L][C][A] (batch-major <-> time-major around the LSTM; the transposes of py/main14b_2.py:156,:166)"""

    @staticmethod
    def forward(ctx, x):
        x = ops._chk(x, "sequence", 3)
        A, C, L = x.shape
        y = _f32(L, C, A, device=x.device)
        lib.wm_permute_acl(_p(x), _p(y), A, C, L, _stream())
        return y

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        L, C, A = g.shape
        y = _f32(A, C, L, device=g.device)
        lib.wm_permute_acl(_p(g), _p(y), L, C, A, _stream())
        return y


class RowsGatherFn(torch.autograd.Function):
    """nn.Embedding lookup (py/main14b_2.py:160) for any width; dense gradient"""

    @staticmethod
    def forward(ctx, table, idx):
        if ops._CHECK_INDEX["mode"] != "off":                  # nn.Embedding raises IndexError (py/main14b_2.py:160)
            ops._note_index_error(((idx < 0) | (idx >= table.shape[0])).any().to(torch.int32).reshape(1), table.shape[0])
            idx = idx.clamp(0, table.shape[0] - 1)              # keep the gather in range whatever the mode reports later
        ctx.save_for_backward(idx)
        ctx.shape = table.shape
        return table.index_select(0, idx).contiguous()          # row gather: data movement only

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        dt = torch.zeros(ctx.shape, dtype=torch.float32, device=g.device)
        lib.wm_rows_scatter_add(_p(dt), _p(idx), _p(g.contiguous()), idx.shape[0], ctx.shape[1], ctx.shape[0], _stream())
        return dt, None


class LSTMLayerFn(ops.GradAwareFunction):
    """one layer of nn.LSTM(H, H) on a time-major sequence [T][H][B] (zero initial state): the input projection of all steps
    is one GEMM, the recurrence a chain of T per-step launches issued by the C launcher (wm_lstm_seq_fwd / _bwd)"""

    @staticmethod
    def forward(ctx, seq, w_ih, w_hh, b_ih, b_hh):
        seq = ops._chk(seq, "sequence", 3)
        T, H, B = seq.shape
        dev, st = seq.device, _stream()
        bias = (b_ih + b_hh).contiguous()
        xp = _gconv_raw(seq, w_ih.t().contiguous(), bias, 1, 1, 0, 4 * H, B, 1, 0, 4 * H, B)     # [T][4H][B]
        need = ops.wants_grad(ctx)
        hs = _f32(T + 1, H, B, device=dev)           # hs[t+1] = h_t, hs[0] = 0 (so hs[:T] is the h_{t-1} sequence)
        cs = _f32(T + 1, H, B, device=dev)
        hs[0].zero_(); cs[0].zero_()
        lib.wm_lstm_seq_fwd(_p(xp), _p(w_hh.contiguous()), _p(hs), _p(cs), T, H, B, 1 if need else 0, st)
        if need:
            ctx.save_for_backward(seq, hs, cs, xp, w_ih, w_hh)      # xp now holds the gate activations
        return hs[1:]

    @staticmethod
    def backward(ctx, dout):
        seq, hs, cs, gates, w_ih, w_hh = ctx.saved_tensors
        ops._single_backward(ctx, "LSTMLayerFn")
        dout = dout.contiguous()
        T, H, B = seq.shape
        dev, st = seq.device, _stream()
        dc = _f32(H, B, device=dev)
        lib.wm_lstm_seq_bwd(_p(gates), _p(cs), _p(dout), _p(w_hh.t().contiguous()), _p(dc), T, H, B, st)   # gates -> da
        da = gates                                   # [T][4H][B]
        dx = _gconv_raw(da, w_ih.contiguous(), None, 1, 1, 0, H, B, 1, 0, H, B, grad_in=True) if ctx.needs_input_grad[0] else None
        dwi, db = _gwgrad_raw(da, seq, H, B, 0, (4 * H, H, 1), 1, 0, True)
        dwh, _ = _gwgrad_raw(da, hs[:T], H, B, 0, (4 * H, H, 1), 1, 0, False)
        return dx, dwi.reshape(4 * H, H), dwh.reshape(4 * H, H), db, db.clone()


# ------------------------------------------------------------------------------------------ modules
def make_conv1d(in_ch, out_ch, kernel_size=3, stride=1, padding=1):
    return nn.Conv1d(in_ch, out_ch, kernel_size, stride=stride, padding=padding)


def _conv(x, m, act=0, res=None, vec=None, grad_rows=None):
    return ConvFn.apply(x, m.weight, m.bias, m.stride[0], m.padding[0], act, res, vec, grad_rows)


class ResidualBlockFn(torch.autograd.Function):
    """ResidualBlock.forward, py/main14b_2.py:95-102: elu(conv2(elu(conv1(x))) + skip(x)) as one tape node.  Backward keeps
    every gradient sum inside a convolution epilogue: conv2's data gradient leaves its kernel already multiplied by
    ELU'(out1), and the two paths into x (conv1 / skip or identity) are summed by the second launch's `res` input."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, ws, bs, stride):
        x = ops._chk(x, "input", 3)
        out1 = _conv_fwd(x, w1, b1, stride, 1, 1)
        res = _conv_fwd(x, ws, bs, stride, 0) if ws is not None else x
        y = _conv_fwd(out1, w2, b2, 1, 1, 1, res)
        ctx.stride = stride
        ctx.has_skip = ws is not None
        ctx.save_for_backward(x, out1, y, w1, w2, ws if ws is not None else x.new_empty(0))
        return y

    @staticmethod
    def backward(ctx, gy):
        x, out1, y, w1, w2, ws = ctx.saved_tensors
        stride = ctx.stride
        gz2 = _elu_bwd(gy.contiguous(), y)                                      # dL/d(conv2 + res)
        dw2, db2 = _conv_wgrad(gz2, out1, w2.shape, 1, 1, True)
        gz1 = _conv_dgrad(gz2, w2, 1, 1, out1.shape[2], res=out1, act=2)        # dL/d(conv1 output before its ELU)
        planes = None
        if stride > 1:                                                          # tap planes of x: shared by conv1 and the skip conv
            planes = _gather_taps(x, 3, stride, 1, gz1.shape[2], 0)
        dw1, db1 = _conv_wgrad(gz1, x, w1.shape, stride, 1, True, planes)
        dws = dbs = None
        if ctx.has_skip:
            Cin, Lo = x.shape[1], gz2.shape[2]
            if stride > 1:                                                      # x[t*stride] is tap plane 1 (padding 1) of conv1's planes
                dws, dbs = _gwgrad_raw(gz2, planes[:, Cin:2 * Cin], Cin, Lo, 3 * Cin * Lo, ws.shape, 1, 0, True)
            else:
                dws, dbs = _conv_wgrad(gz2, x, ws.shape, 1, 0, True)
        dx = None
        if ctx.needs_input_grad[0]:
            if ctx.has_skip and stride >= 3 and _FUSED_STRIDED_DGRAD:
                dx = _strided_block_dgrad(gz1, w1, gz2, ws, stride, x.shape[2])
            else:
                via_skip = _conv_dgrad(gz2, ws, stride, 0, x.shape[2]) if ctx.has_skip else gz2
                dx = _conv_dgrad(gz1, w1, stride, 1, x.shape[2], res=via_skip)
        return dx, dw1, db1, dw2, db2, dws, dbs, None


class ResidualBlock(nn.Module):
    """py/main14b_2.py:86-102"""

    def __init__(self, in_ch, out_ch, stride=1):
        super().__init__()
        self.downsample = (stride != 1 or in_ch != out_ch)
        self.conv1 = make_conv1d(in_ch, out_ch, kernel_size=3, stride=stride, padding=1)
        self.conv2 = make_conv1d(out_ch, out_ch, kernel_size=3, stride=1, padding=1)
        self.elu = nn.ELU()
        if self.downsample:
            self.skip_conv = make_conv1d(in_ch, out_ch, kernel_size=1, stride=stride, padding=0)

    def forward(self, x):
        sk = self.skip_conv if self.downsample else None
        return ResidualBlockFn.apply(x, self.conv1.weight, self.conv1.bias, self.conv2.weight, self.conv2.bias,
                                     sk.weight if sk is not None else None, sk.bias if sk is not None else None,
                                     self.conv1.stride[0])


def _fit_length(y, T):
    """py/main14b_2.py:172-177 / :217-222 (slicing and zero padding: data movement)"""
    if y.shape[-1] > T:
        y = y[:, :, :T]
    elif y.shape[-1] < T:
        y = torch.nn.functional.pad(y, (0, T - y.shape[-1]))
    return y.contiguous()


class Generator(nn.Module):
    """py/main14b_2.py:104-177"""

    def __init__(self, in_channels=1, base_channels=CHANNELS, hidden_dim=HIDDEN_DIM, message_bits=NUM_BITS,
                 output_channels=OUTPUT_CH, strides=STRIDES):
        super().__init__()
        if hidden_dim % 32 != 0 or not 32 <= hidden_dim <= 256:
            raise ValueError(f"hidden_dim must be a multiple of 32 in [32, 256] (got {hidden_dim}): the recurrence kernels "
                             "(wm_lstm_seq_fwd/bwd) keep one gate row of W_hh per lane in registers, sized for H <= 256")
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
        s = ops._chk(s, "clip batch", 3)
        T = s.shape[-1]
        x = _conv(s, self.init_conv)
        x = self.encoder_blocks(x)                                                   # (B,512,T/320)
        vec = RowsGatherFn.apply(self.E.weight, message.to(torch.int64)) if message is not None else None
        xt = ConvFn.apply(x, self.proj.weight.unsqueeze(-1), self.proj.bias, 1, 0, 0, None, vec, None)   # proj + embedding add
        seq = PermuteFn.apply(xt)                                                    # [T'][hd][B]
        for l in range(self.lstm.num_layers):
            seq = LSTMLayerFn.apply(seq, getattr(self.lstm, f"weight_ih_l{l}"), getattr(self.lstm, f"weight_hh_l{l}"),
                                    getattr(self.lstm, f"bias_ih_l{l}"), getattr(self.lstm, f"bias_hh_l{l}"))
        x = _conv(PermuteFn.apply(seq), self.final_conv_enc)                          # (B,128,T')
        for i, st in enumerate(reversed(self.strides)):
            ct = self.decoder_blocks[2 * i]
            x = ConvTFn.apply(x, ct.weight, ct.bias, st)
            x = self.decoder_blocks[2 * i + 1](x)
        return _fit_length(_conv(x, self.final_conv_dec), T)


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

    def forward(self, x, input_grad_rows=None):
        """`input_grad_rows` (optional, not in the reference): only clips [0, input_grad_rows) get an input gradient -- the
        train step feeds [watermarked; clean] (py/main14b_2.py:306) and the clean half is data"""
        x = ops._chk(x, "clip batch", 3)
        T = x.shape[-1]
        y = _conv(x, self.init_conv, grad_rows=input_grad_rows)
        y = self.encoder_blocks(y)
        for i, st in enumerate(reversed(self.strides)):
            ct = self.upsample_blocks[2 * i]
            y = ConvTFn.apply(y, ct.weight, ct.bias, st)
            y = self.upsample_blocks[2 * i + 1](y)
        return _fit_length(_conv(y, self.final_conv), T)


# kept for the forward-only unit tests
def _gconv(x, w, bias=None, stride=1, padding=0, act=0, res=None, vec=None):
    return _conv_fwd(ops._chk(x, "input", 3), w, bias, stride, padding, act, res, vec)


def _gconvT(x, w, bias, st):
    with torch.no_grad():
        return ConvTFn.apply(x, w, bias, st)


# ------------------------------------------------------------------------------------------ step recipe
LOSS_WEIGHTS = dict(l1=0.1, mel=2.0, loud=10.0, loc=10.0, bce=1.0)      # py/main14b_2.py:34-38


def forward_losses(generator, detector, s, message):
    """loop body of train_one_epoch, py/main14b_2.py:300-352: s_w is clamped to +-1 (:305), logits are channel-first,
    no FIR / RMS / HF terms"""
    from collections import OrderedDict
    from . import losses as L
    delta = generator(s, message)
    s_w = L.clamp_peak(s + delta, 1.0)                                   # torch.clamp(s_w, -1, 1)
    logits = detector(torch.cat([s_w, s], dim=0), input_grad_rows=s.shape[0])   # (2B, 1+bits, T)
    loc, bce = L.detection_losses(logits.permute(0, 2, 1).contiguous(), message)
    l1 = L.l1_to_zero(delta)
    mel = L.MultiScaleMelLoss()(s, s_w)
    loud = L.TFLoudnessLoss()(s, s_w)
    w = LOSS_WEIGHTS
    raw = l1 + mel + loud + loc + bce
    total = w["l1"] * l1 + w["mel"] * mel + w["loud"] * loud + w["loc"] * loc + w["bce"] * bce
    return total, OrderedDict(delta=delta, s_w=s_w, logits=logits, l1=l1, mel=mel, loud=loud, loc=loc, bce=bce,
                              raw_total=raw, total=total)


def train_step(generator, detector, optimizer, s, message, grad_sync=None):
    optimizer.zero_grad(set_to_none=not hasattr(optimizer, "flat"))
    if hasattr(grad_sync, "begin_step"):
        grad_sync.begin_step()
    try:
        with ops.index_check_mode("deferred" if ops._CHECK_INDEX["mode"] == "sync" else ops._CHECK_INDEX["mode"]):
            total, out = forward_losses(generator, detector, s, message)   # no mid-step sync (it would drain the launch queue)
        total.backward()
        if hasattr(optimizer, "finish_backward"):
            optimizer.finish_backward()
        if grad_sync is not None:
            grad_sync()
        ops.check_message_ids(wait=True, what="this train_step's batch")    # a bad id raises before the update (step.train_step)
    except BaseException:
        ops.drop_pending_message_checks()        # a step that died half-way must not report its flag inside a later, valid step
        raise
    optimizer.step()
    return out
