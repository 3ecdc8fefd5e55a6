"""torch.autograd.Function wrappers over the C ABI (include/wm_hip.h).

PyTorch is used here for device memory, streams and the autograd tape only; every
arithmetic step of the hot path is a launch into libwm_hip.so.  Each Function mirrors one
block of the reference graph (py/main16.py:112-186, :53-81, :192-217).
"""
from __future__ import annotations

import torch

from ._lib import lib

NCU = 256            # workgroups the persistent kernels are sized for (MI355X CU count)
BN_EPS = 1e-5        # nn.BatchNorm1d defaults (py/main16.py:117,120)
BN_MOMENTUM = 0.1


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def _chk(t: torch.Tensor, name: str, ndim: int | None = None, dtype=torch.float32) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: the watermark hot path runs on the GPU only (got a {t.device} tensor); "
                           "there is no CPU fallback")
    if t.dtype != dtype:
        raise ValueError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if ndim is not None and t.dim() != ndim:
        raise ValueError(f"{name}: expected {ndim} dims, got shape {tuple(t.shape)}")
    return t.contiguous()


def _frames(x: torch.Tensor, name: str, ch: int) -> torch.Tensor:
    x = _chk(x, name, 3)
    if x.shape[1] != ch:
        raise ValueError(f"{name}: expected {ch} channels, got shape {tuple(x.shape)}")
    if x.shape[2] % 4 != 0:
        raise ValueError(f"{name}: clip length must be a multiple of 4 samples, got {x.shape[2]}")
    return x


def _f32(*shape, device):
    return torch.empty(shape, dtype=torch.float32, device=device)


# ---------------------------------------------------------------------------------------------- grad mode
# Inside Function.forward autograd has already switched grad mode off, and ctx.needs_input_grad only mirrors the
# inputs' requires_grad flags -- under torch.no_grad() a module with trainable parameters still reports True there.
# The forward-only fast paths (two-launch inference ResBlock, LSTM without saved activations) must key on "will a
# backward ever run", i.e. on the grad mode of the CALLER: GradAwareFunction.apply records it before dispatch.
_GRAD = {"on": True}


class GradAwareFunction(torch.autograd.Function):
    @classmethod
    def apply(cls, *args, **kwargs):
        _GRAD["on"] = torch.is_grad_enabled()
        return super().apply(*args, **kwargs)


def wants_grad(ctx) -> bool:
    """True when the tape is recording AND some input asks for a gradient"""
    return _GRAD["on"] and any(ctx.needs_input_grad)


def _single_backward(ctx, what):
    """the recurrences overwrite their saved gate activations with da in place (no second 4-5 GB buffer at B=256):
    a second backward over a retained graph would read da as if it were activations -- refuse it loudly"""
    if getattr(ctx, "_wm_consumed", False):
        raise RuntimeError(f"{what}: backward was already run once on this graph; the saved gate activations were "
                           "overwritten in place (retain_graph=True / per-loss backward calls are not supported -- "
                           "sum the losses and call backward once, as py/main16.py:275-277 does)")
    ctx._wm_consumed = True


# ---------------------------------------------------------------------------------------------- side stream
# Weight-gradient GEMMs do not feed the data path of backward.  When the parameters carry a pre-allocated gradient
# destination (optim.FlatAdam sets p._wm_grad = view of the flat gradient bucket and enables this), they are
# launched on a second HIP stream and ACCUMULATE straight into that bucket (the Function then returns None for
# them), so they overlap with the latency-bound LSTM recurrence and the HBM-bound element-wise kernels of the main
# stream.  optim.FlatAdam.finish_backward() joins the streams before the all-reduce / update.
_ASYNC = {"on": False, "side": None, "deferred": []}


def set_async_wgrad(on: bool):
    _ASYNC["on"] = bool(on)


def side_stream():
    if _ASYNC["side"] is None:
        _ASYNC["side"] = torch.cuda.Stream()
    return _ASYNC["side"]


def release_deferred_wgrads():
    """Weight-gradient GEMMs queued so far start now, on the side stream, behind everything the current stream has been
    given up to this point.  Called right before the LSTM BPTT launch (the 8 ms during which the matrix cores are idle and
    each CU holds only one 152-register recurrence workgroup) and from join_side_stream()."""
    pend, _ASYNC["deferred"] = _ASYNC["deferred"], []
    if not pend:
        return
    main, side = torch.cuda.current_stream(), side_stream()
    ev = torch.cuda.Event()
    ev.record(main)
    with torch.cuda.stream(side):
        side.wait_event(ev)
        for inputs, fn in pend:
            for t in inputs:
                if t is not None:
                    t.record_stream(side)
            fn()


def join_side_stream():
    release_deferred_wgrads()
    if _ASYNC["side"] is not None:
        torch.cuda.current_stream().wait_stream(_ASYNC["side"])


def _gdst(*params):
    """gradient destinations registered on the parameters (None when the async path is off)"""
    if not _ASYNC["on"]:
        return tuple(None for _ in params)
    return tuple(getattr(p, "_wm_grad", None) for p in params)


def _on_side(inputs, fn, defer=True):
    """queue fn() (kernel launches) for the side stream.  defer=True: held back until release_deferred_wgrads() -- launched
    at once they would only take turns with the main stream's convolutions (both want whole CUs); released at the start of
    the LSTM BPTT they fill the CUs' idle matrix cores instead."""
    if defer:
        _ASYNC["deferred"].append((inputs, fn))
        return
    _ASYNC["deferred"].append((inputs, fn))
    release_deferred_wgrads()


# ---------------------------------------------------------------------------------------------- conv arithmetic mode
# k3 convolutions (forward + data gradient): native fp32 MFMA, or the bf16x6 split build (fp32-grade error on the
# bf16 matrix cores, see csrc/conv64.hip).  WM_CONV_BF16X6=0/1 in the environment overrides the default.
import os as _os
_CONV = {"bf16x6": _os.environ.get("WM_CONV_BF16X6", "1") == "1", "schedule": 2,
         "one_launch_eval": _os.environ.get("WM_RESBLOCK_ONE_LAUNCH", "1") == "1",
         "fused_bwd": _os.environ.get("WM_FUSED_BWD", "1") == "1",
         "mask_on_load": _os.environ.get("WM_MASK_ON_LOAD", "1") == "1",
         "pair_fold": _os.environ.get("WM_PAIR_FOLD", "1") == "1",
         "bwd_f16x3": _os.environ.get("WM_BWD_F16X3", "1") == "1",
         "fwd_f16x3": _os.environ.get("WM_FWD_F16X3", "1") == "1",
         "conv7_f16x3": _os.environ.get("WM_CONV7_F16X3", "1") == "1",
         "eval_f16x3": _os.environ.get("WM_EVAL_F16X3", "1") == "1"}


def set_conv_bf_schedule(schedule: int):
    """0: phase-serial kernel, 2: register-resident weights + interleaved split (csrc/conv64.hip)."""
    lib.wm_set_conv_bf_schedule(int(schedule), None)
    _CONV["schedule"] = int(schedule)


if "WM_CONV_BF_SCHEDULE" in _os.environ:
    set_conv_bf_schedule(int(_os.environ["WM_CONV_BF_SCHEDULE"]))


def set_conv_bf16x6(on: bool):
    _CONV["bf16x6"] = bool(on)


def set_fused_backward(on: bool):
    """ResBlock backward: data gradient + weight gradient of each convolution in ONE launch (wm_dwgrad64_bf, default) or as two
    (wm_conv64_bf + wm_wgrad64_bf).  WM_FUSED_BWD=0/1 in the environment sets the default."""
    _CONV["fused_bwd"] = bool(on)


def set_mask_on_load(on: bool):
    """fused ResBlock backward: 1 (default) the masked gradient dz2 = g * (out > 0) is never written -- the reduction pass forms
    only the BatchNorm sums and wm_dwgrad64_bf masks g on load; 0 it is materialised first.  WM_MASK_ON_LOAD=0/1 sets the default."""
    _CONV["mask_on_load"] = bool(on)


def set_bwd_f16x3(on: bool):
    """fused ResBlock backward arithmetic: 1 (default) f16 two-piece split, three products per product on the f16 matrix cores
    (half the matrix work of bf16x6; power-of-two scales from max |w| and max |A| max |dz| keep the operands in the f16 range),
    0 bf16x6 as in the forward.  WM_BWD_F16X3=0/1 sets the default."""
    _CONV["bwd_f16x3"] = bool(on)


def set_fwd_f16x3(on: bool):
    """ResBlock forward convolutions launched through wm_conv64_bf (schedule 2, T % 128 == 0): 1 (default) the f16 two-piece split
    (three products per product instead of six; weights scaled by a power of two from max |w|, activations unscaled), 0 bf16x6.
    WM_FWD_F16X3=0/1 sets the default.  The one-launch inference ResBlock has its own switch (set_eval_f16x3)."""
    _CONV["fwd_f16x3"] = bool(on)


def set_eval_f16x3(on: bool):
    """one-launch inference ResBlock (wm_resblock_eval_bf): 1 (default) the f16 two-piece split, 0 bf16x6.  WM_EVAL_F16X3=0/1."""
    _CONV["eval_f16x3"] = bool(on)


def pack_w64_h(w: torch.Tensor, mode: int) -> torch.Tensor:
    wph = torch.empty(2 * 3 * 4096 + 4, dtype=torch.int16, device=w.device)      # two f16 pieces + {ws, 1 / ws}
    lib.wm_pack_w64_h(_p(w), _p(wph), mode, _stream())
    return wph


def set_pair_fold(on: bool):
    """two ResBlocks in a row (encoder.1 -> encoder.2, model.1 -> model.2) as ONE tape node whose backward lets the second block's
    conv1 launch also do the first block's ReLU backward and BatchNorm sums (no reduction pass for the first block).
    WM_PAIR_FOLD=0/1 sets the default; off = two ResBlockFn nodes."""
    _CONV["pair_fold"] = bool(on)


def set_resblock_one_launch(on: bool):
    """inference ResBlock: one launch (wm_resblock_eval_bf, default) or the two-launch form (conv1; conv2 with BN2 + add + ReLU
    in its epilogue).  WM_RESBLOCK_ONE_LAUNCH=0/1 in the environment sets the default."""
    _CONV["one_launch_eval"] = bool(on)


def conv_bf16x6() -> bool:
    return _CONV["bf16x6"]


def pack_w64_bf(w: torch.Tensor, mode: int) -> torch.Tensor:
    wpb = torch.empty(3 * 3 * 4096, dtype=torch.int16, device=w.device)
    lib.wm_pack_w64_bf(_p(w), _p(wpb), mode, _stream())
    return wpb


def set_conv7_f16x3(on: bool):
    """ConvTranspose1d(64,64,7) forward / data gradient / weight gradient (T % 128 == 0): 1 (default) the f16 two-piece split (three
    products per product; weights scaled by a power of two from max |w|, the incoming gradient by one from max |g| -- one streaming
    pass over g, wm_gscale_absmax), 0 bf16x6.  WM_CONV7_F16X3=0/1 sets the default."""
    _CONV["conv7_f16x3"] = bool(on)


def pack_w64_h7(w: torch.Tensor, mode: int) -> torch.Tensor:
    wph = torch.empty(2 * 7 * 4096 + 4, dtype=torch.int16, device=w.device)      # two f16 pieces + {ws, 1 / ws}
    lib.wm_pack_w64_h7(_p(w), _p(wph), mode, _stream())
    return wph


# max |g| per producer workgroup of gradient tensors whose producer formed it anyway, keyed by the tensor OBJECT (a weak reference
# guards against a recycled id): a hit saves the consumer's streaming pass over g, a miss is merely slower
_GMAX = {}


def _note_gmax(t: torch.Tensor, maxes: torch.Tensor):
    import weakref
    if len(_GMAX) > 64:
        _GMAX.clear()
    _GMAX[id(t)] = (weakref.ref(t), maxes)


def gscale_of(g: torch.Tensor, log2_target: float = 12.0) -> torch.Tensor:
    """{gs, 1 / gs} for gradient g: from its producer's per-workgroup maxima when it left them (no pass over g), else wm_gscale_absmax"""
    e = _GMAX.pop(id(g), None)
    if e is not None and e[0]() is g:
        gsc = _f32(2, device=g.device)
        lib.wm_gscale_from_max(_p(e[1]), e[1].numel(), float(log2_target), _p(gsc), _stream())
        return gsc
    return gscale_absmax(g, log2_target)


def gscale_absmax(g: torch.Tensor, log2_target: float = 12.0) -> torch.Tensor:
    """{gs, 1 / gs}: the power of two that puts max |g| into (2^(L-1), 2^L] -- the input scale of the f16 two-piece split kernels"""
    gsc, scratch = _f32(2, device=g.device), _f32(1024, device=g.device)
    lib.wm_gscale_absmax(_p(g), g.numel(), _p(scratch), float(log2_target), _p(gsc), _stream())
    return gsc


def pack_w64_bf7(w: torch.Tensor, mode: int) -> torch.Tensor:
    wpb = torch.empty(3 * 7 * 4096, dtype=torch.int16, device=w.device)
    lib.wm_pack_w64_bf7(_p(w), _p(wpb), mode, _stream())
    return wpb


def _conv3(x, x2, w, mode, pa, pb, pc, bias, e1, ea, eb, y, stats, B, T, pro, epi):
    """one k3 64->64 convolution launch in the selected arithmetic mode (mode: 0 forward, 1 data gradient)"""
    if _CONV["bf16x6"]:
        h = (_CONV["fwd_f16x3"] and mode == 0 and (pro, epi) in ((0, 0), (1, 0), (1, 4)) and _CONV["schedule"] == 2 and T % 128 == 0)
        lib.wm_conv64_bf(_p(x), _p(x2), _p(pack_w64_h(w, 0) if h else pack_w64_bf(w, mode)), _p(pa), _p(pb), _p(pc), _p(bias), _p(e1),
                         _p(ea), _p(eb), _p(y), _p(stats), B, T, pro, epi, 1 if h else 0, _stream())
    else:
        lib.wm_conv64(_p(x), _p(x2), _p(pack_w64(w, 3, mode)), _p(pa), _p(pb), _p(pc), _p(bias), _p(e1), _p(ea), _p(eb), _p(y),
                      _p(stats), B, T, 3, pro, epi, _stream())


def pack_w64(w: torch.Tensor, kw: int, mode: int) -> torch.Tensor:
    wp = _f32(kw * 4096, device=w.device)
    lib.wm_pack_w64(_p(w), _p(wp), kw, mode, _stream())
    return wp


# ------------------------------------------------------------------------------------------ ResBlock
class ResBlockFn(GradAwareFunction):
    """relu(x + BN2(conv2(relu(BN1(conv1(x))))))  -- ResBlock.forward, py/main16.py:124-125."""

    @staticmethod
    def forward(ctx, x, w1, b1, g1, be1, w2, b2, g2, be2, rm1, rv1, nbt1, rm2, rv2, nbt2, training):
        x = _frames(x, "ResBlock input", 64)
        B, _, T = x.shape
        dev, st = x.device, _stream()
        out = torch.empty_like(x)
        cst = _f32(8, 64, device=dev)       # sc1 sh1 mean1 is1 sc2 sh2 mean2 is2
        sc1, sh1, mu1, is1, sc2, sh2, mu2, is2 = cst.unbind(0)
        if training:
            y1, y2 = torch.empty_like(x), torch.empty_like(x)
            stats = _f32(NCU * 128, device=dev)
            _conv3(x, None, w1, 0, None, None, None, b1, None, None, None, y1, stats, B, T, 0, 0)
            lib.wm_bn_finalize(_p(stats), NCU, float(B * T), _p(g1), _p(be1), _p(rm1), _p(rv1), _p(nbt1), BN_MOMENTUM, BN_EPS,
                               _p(sc1), _p(sh1), _p(mu1), _p(is1), st)
            _conv3(y1, None, w2, 0, sc1, sh1, None, b2, None, None, None, y2, stats, B, T, 1, 0)
            lib.wm_bn_finalize(_p(stats), NCU, float(B * T), _p(g2), _p(be2), _p(rm2), _p(rv2), _p(nbt2), BN_MOMENTUM, BN_EPS,
                               _p(sc2), _p(sh2), _p(mu2), _p(is2), st)
        else:
            lib.wm_bn_eval_scale_shift(_p(g1), _p(be1), _p(rm1), _p(rv1), BN_EPS, _p(sc1), _p(sh1), st)
            lib.wm_bn_eval_scale_shift(_p(g2), _p(be2), _p(rm2), _p(rv2), BN_EPS, _p(sc2), _p(sh2), st)
            if not wants_grad(ctx) and _CONV["bf16x6"] and _CONV["one_launch_eval"]:
                # inference: the whole block is ONE launch -- x in, out out, the intermediate activation stays in LDS
                h = _CONV["eval_f16x3"]                     # f16 two-piece split (three products per product) | bf16x6
                wp1 = torch.empty((2 * 3 * 4096 + 4) if h else 3 * 3 * 4096, dtype=torch.int16, device=dev)   # both images alive at the launch
                wp2 = torch.empty_like(wp1)
                pack = lib.wm_pack_w64_h_scaled if h else lib.wm_pack_w64_bf_scaled
                pack(_p(w1), _p(sc1), _p(wp1), st)
                pack(_p(w2), _p(sc2), _p(wp2), st)
                lib.wm_resblock_eval_bf(_p(x), _p(wp1), _p(wp2), _p(b1), _p(sc1), _p(sh1), _p(b2), _p(sc2), _p(sh2), _p(out), B, T,
                                        1 if h else 0, st)
                return out
            y1 = torch.empty_like(x)
            _conv3(x, None, w1, 0, None, None, None, b1, None, None, None, y1, None, B, T, 0, 0)
            if not wants_grad(ctx) and _CONV["bf16x6"] and _CONV["schedule"] == 2 and T % 128 == 0:
                # inference: BN2 + residual add + ReLU ride in conv2's epilogue -- the block is two launches
                _conv3(y1, None, w2, 0, sc1, sh1, None, b2, x, sc2, sh2, out, None, B, T, 1, 4)
                return out
            y2 = torch.empty_like(x)
            _conv3(y1, None, w2, 0, sc1, sh1, None, b2, None, None, None, y2, None, B, T, 1, 0)
            # saved (mean, invstd) for an eval-mode backward = running statistics
            mu1.copy_(rm1); is1.copy_(torch.rsqrt(rv1 + BN_EPS)); mu2.copy_(rm2); is2.copy_(torch.rsqrt(rv2 + BN_EPS))
        if wants_grad(ctx):
            # the backward needs only the SIGN of `out`: one bit per element, written beside it (a frame pass less in backward)
            mask = torch.empty(B * 64 * ((T + 31) // 32), dtype=torch.int32, device=dev)
            lib.wm_bn_add_relu_mask(_p(x), _p(y2), _p(sc2), _p(sh2), _p(out), _p(mask), B, T, st)
        else:
            mask = None
            lib.wm_bn_add_relu(_p(x), _p(y2), _p(sc2), _p(sh2), _p(out), B, T, st)
        ctx.training = bool(training)
        ctx.gdst = _gdst(w1, b1, w2, b2)
        ctx._wm_saved = (x, y1, y2, mask, cst, w1, w2, g1, g2)
        if not getattr(ctx, "_wm_pair", False):
            ctx.save_for_backward(*ctx._wm_saved)
        return out

    @staticmethod
    def backward(ctx, g_out):
        x, y1, y2, mask, cst, w1, w2, g1, g2 = ctx.saved_tensors
        sc1, sh1, mu1, is1, sc2, sh2, mu2, is2 = cst.unbind(0)
        g_out = g_out.contiguous()
        B, _, T = x.shape
        dev, st = x.device, _stream()
        ev = 0 if ctx.training else 1
        n = float(B * T)
        gw1, gb1, gw2, gb2 = ctx.gdst
        side = all(g is not None for g in ctx.gdst)
        fused = _CONV["bf16x6"] and _CONV["fused_bwd"] and not side and T % 64 == 0
        if fused:
            dx, grads, _ = _resblock_bwd_fused(ctx.saved_tensors, ctx.training, g_out)
            return (dx,) + grads + (None,) * 7
        part = _f32(max(B, 1) * 128, device=dev)
        dz2 = torch.empty_like(x)
        lib.wm_relu_bwd_reduce_mask(_p(g_out), _p(mask), _p(y2), _p(dz2), _p(part), None, B, T, st)
        k2 = _f32(4, 64, device=dev)          # A, B (hi), B (lo), C  -- B is handed over as hi + lo words
        dg2, dbe2 = _f32(64, device=dev), _f32(64, device=dev)
        lib.wm_bn_bwd_finalize(_p(part), B, n, _p(g2), _p(mu2), _p(is2), _p(k2[0]), _p(k2[1]), _p(k2[3]), _p(dg2), _p(dbe2), 0, ev, None, 0, None, st)
        # conv2: data gradient (+ ReLU mask + BN1-backward reductions in the epilogue) and weight gradient
        dz1 = torch.empty_like(x)
        stats = _f32(NCU * 128, device=dev)
        _conv3(dz2, y2, w2, 1, k2[0], k2[1], k2[3], None, y1, sc1, sh1, dz1, stats, B, T, 3, 1)

        def wgrad2():
            wpart = _f32(2 * NCU * (3 * 4096 + 64), device=dev)
            if _CONV["bf16x6"]:
                lib.wm_wgrad64_bf(_p(dz2), _p(y2), _p(k2[0]), _p(k2[1]), _p(k2[3]), _p(y1), _p(sc1), _p(sh1), _p(wpart),
                                  _p(gw2 if side else dw2), _p(gb2 if side else db2), B, T, 3, 1, 3 if side else 0, _stream())
            else:
                lib.wm_wgrad64(_p(dz2), _p(y2), _p(k2[0]), _p(k2[1]), _p(k2[3]), _p(y1), _p(sc1), _p(sh1), _p(wpart),
                               _p(gw2 if side else dw2), _p(gb2 if side else db2), B, T, 3, 3, 1, 0, 1 if side else 0, _stream())
        dw2, db2 = (None, None) if side else (torch.empty_like(w2), _f32(64, device=dev))
        if side:
            _on_side((dz2, y2, k2, y1, cst), wgrad2)
        else:
            wgrad2()
        k1 = _f32(4, 64, device=dev)
        dg1, dbe1 = _f32(64, device=dev), _f32(64, device=dev)
        lib.wm_bn_bwd_finalize(_p(stats), NCU, n, _p(g1), _p(mu1), _p(is1), _p(k1[0]), _p(k1[1]), _p(k1[3]), _p(dg1), _p(dbe1), 0, ev, None, 0, None, st)
        # conv1: data gradient + residual path, weight gradient
        dx = torch.empty_like(x)
        _conv3(dz1, y1, w1, 1, k1[0], k1[1], k1[3], None, dz2, None, None, dx, None, B, T, 3, 2)
        def wgrad1():
            wpart = _f32(2 * NCU * (3 * 4096 + 64), device=dev)
            if _CONV["bf16x6"]:
                lib.wm_wgrad64_bf(_p(dz1), _p(y1), _p(k1[0]), _p(k1[1]), _p(k1[3]), _p(x), None, None, _p(wpart),
                                  _p(gw1 if side else dw1), _p(gb1 if side else db1), B, T, 3, 0, 3 if side else 0, _stream())
            else:
                lib.wm_wgrad64(_p(dz1), _p(y1), _p(k1[0]), _p(k1[1]), _p(k1[3]), _p(x), None, None, _p(wpart),
                               _p(gw1 if side else dw1), _p(gb1 if side else db1), B, T, 3, 3, 0, 0, 1 if side else 0, _stream())
        dw1, db1 = (None, None) if side else (torch.empty_like(w1), _f32(64, device=dev))
        if side:
            _on_side((dz1, y1, k1, x), wgrad1)
        else:
            wgrad1()
        return dx, dw1, db1, dg1, dbe1, dw2, db2, dg2, dbe2, None, None, None, None, None, None, None


def _resblock_bwd_fused(saved, training, g_out, pre=None, fold=None):
    """Backward of one ResBlock on the fused path (data + weight gradient of each convolution in one launch, T % 64 == 0).
    g_out: gradient w.r.t. the block output.  `pre` = (stats partials [NCU,2,64], max |dz| per workgroup [NCU]) when g_out ALREADY is
    dz2 = g (out > 0) and its two BatchNorm sums exist (made by the next block's folded conv1 launch): no reduction pass, no mask.
    `fold` = (mask, y2) of the block BEFORE this one: the conv1 launch then writes that block's dz2 instead of the plain input
    gradient and returns (its BatchNorm-sum partials, its max |dz| per workgroup) as the third value.
    Returns (dx, (dw1, db1, dg1, dbe1, dw2, db2, dg2, dbe2), fold outputs | None)."""
    x, y1, y2, mask, cst, w1, w2, g1, g2 = saved
    sc1, sh1, mu1, is1, sc2, sh2, mu2, is2 = cst.unbind(0)
    B, _, T = x.shape
    dev, st = x.device, _stream()
    ev = 0 if training else 1
    n = float(B * T)
    h = 1 if _CONV["bwd_f16x3"] else 0        # arithmetic of the two launches; the f16 split needs the gradient's scale (max |dz|)
    pack = pack_w64_h if h else pack_w64_bf
    k2 = _f32(4, 64, device=dev)          # A, B (hi), B (lo), C  -- B is handed over as hi + lo words
    dg2, dbe2 = _f32(64, device=dev), _f32(64, device=dev)
    gs2 = _f32(2, device=dev) if h else None
    if pre is not None:
        ppart, pmax = pre
        gsrc, gm = g_out, None
        lib.wm_bn_bwd_finalize(_p(ppart), NCU, n, _p(g2), _p(mu2), _p(is2), _p(k2[0]), _p(k2[1]), _p(k2[3]), _p(dg2), _p(dbe2), 0, ev,
                               _p(pmax) if h else None, NCU, _p(gs2), st)
    else:
        # dz2 = g_out * (out > 0) is never written (default): the reduction pass only forms the two BatchNorm sums, and the two
        # convolution-backward launches mask g_out with the same bits while they load it (wm_dwgrad64_bf's gmask)
        part = _f32(max(B, 1) * 128, device=dev)
        dzm = _f32(B * 64, device=dev) if h else None
        dz2 = None if _CONV["mask_on_load"] else torch.empty_like(x)
        lib.wm_relu_bwd_reduce_mask(_p(g_out), _p(mask), _p(y2), _p(dz2), _p(part), _p(dzm), B, T, st)
        lib.wm_bn_bwd_finalize(_p(part), B, n, _p(g2), _p(mu2), _p(is2), _p(k2[0]), _p(k2[1]), _p(k2[3]), _p(dg2), _p(dbe2), 0, ev,
                               _p(dzm), B * 64, _p(gs2), st)
        gsrc, gm = (g_out, mask) if dz2 is None else (dz2, None)
    # conv2: data gradient (+ ReLU mask + BN1-backward reductions in the epilogue) and weight gradient
    dz1 = torch.empty_like(x)
    stats = _f32(NCU * 128, device=dev)
    dzm1 = _f32(NCU, device=dev) if h else None          # max |dz1| per workgroup, written by the conv2-pair launch
    dw2, db2, dw1, db1 = torch.empty_like(w2), _f32(64, device=dev), torch.empty_like(w1), _f32(64, device=dev)
    wpart = _f32(NCU * (3 * 4096 + 64), device=dev)
    lib.wm_dwgrad64_bf(_p(gsrc), _p(y2), _p(k2[0]), _p(k2[1]), _p(k2[3]), _p(pack(w2, 1)), _p(y1), _p(sc1), _p(sh1),
                       _p(y1), _p(sc1), _p(sh1), _p(dz1), _p(stats), _p(wpart), _p(dw2), _p(db2), B, T, 1, 1, 0, _p(gm), h, _p(gs2), _p(dzm1), st)
    k1 = _f32(4, 64, device=dev)
    dg1, dbe1 = _f32(64, device=dev), _f32(64, device=dev)
    gs1 = _f32(2, device=dev) if h else None
    lib.wm_bn_bwd_finalize(_p(stats), NCU, n, _p(g1), _p(mu1), _p(is1), _p(k1[0]), _p(k1[1]), _p(k1[3]), _p(dg1), _p(dbe1), 0, ev,
                           _p(dzm1), NCU, _p(gs1), st)
    dx = torch.empty_like(x)
    fout = None
    if fold is not None and gm is not None:
        # conv1 pair that also does the PREVIOUS block's ReLU backward and BatchNorm sums (epi 8): dx leaves as that block's dz2
        pmask, py2 = fold
        fpart = _f32(NCU * 128, device=dev)
        fmax = _f32(NCU, device=dev) if h else None
        lib.wm_dwgrad64_bf(_p(dz1), _p(y1), _p(k1[0]), _p(k1[1]), _p(k1[3]), _p(pack(w1, 1)), _p(x), None, None,
                           _p(gsrc), _p(py2), _p(pmask), _p(dx), _p(fpart), _p(wpart), _p(dw1), _p(db1), B, T, 0, 8, 0, _p(gm), h, _p(gs1), _p(fmax), st)
        fout = (fpart, fmax)
    else:
        xmax = _f32(NCU, device=dev) if h else None          # max |dx| per workgroup: a consumer that splits dx into f16 pieces
        lib.wm_dwgrad64_bf(_p(dz1), _p(y1), _p(k1[0]), _p(k1[1]), _p(k1[3]), _p(pack(w1, 1)), _p(x), None, None,
                           _p(gsrc), None, None, _p(dx), None, _p(wpart), _p(dw1), _p(db1), B, T, 0, 2, 0, _p(gm), h, _p(gs1), _p(xmax), st)
        if h:
            _note_gmax(dx, xmax)                             # (ConvT7Fn.backward) then needs no pass over it for its scale
    return dx, (dw1, db1, dg1, dbe1, dw2, db2, dg2, dbe2), fout


class ResBlockPairFn(GradAwareFunction):
    """Two ResBlocks in a row (py/main16.py:135-136 encoder.1 -> encoder.2, :178-179 model.1 -> model.2) as one tape node.
    Forward = ResBlockFn's training forward twice.  Backward: the second block's conv1 launch (data + weight gradient) applies the
    FIRST block's ReLU mask to the gradient it has just formed and accumulates that block's two BatchNorm sums in its epilogue
    (wm_dwgrad64_bf epi 8), so the first block needs no reduction pass and the gradient between the blocks is written once, masked.
    Used by modules.resblock_pair when the fused path applies (bf16x6, T % 64 == 0, mask-on-load, gradients wanted)."""

    @staticmethod
    def forward(ctx, x, *args):
        p1, p2, training = args[:14], args[14:28], args[28]
        ctx._wm_pair = True                                  # ResBlockFn.forward then leaves the saving to this node
        mid = ResBlockFn.forward(ctx, x, *p1, training)
        ctx.saved1 = ctx._wm_saved                       # (x, y1, y2, mask, cst, w1, w2, g1, g2) of block 1
        out = ResBlockFn.forward(ctx, mid, *p2, training)
        ctx.save_for_backward(*(ctx.saved1 + ctx._wm_saved))
        ctx.training = bool(training)
        return out

    @staticmethod
    def backward(ctx, g_out):
        sv = ctx.saved_tensors
        s1_, s2_ = sv[:9], sv[9:]
        g_out = g_out.contiguous()
        dmid, grads2, fout = _resblock_bwd_fused(s2_, ctx.training, g_out, fold=(s1_[3], s1_[2]))
        dx, grads1, _ = _resblock_bwd_fused(s1_, ctx.training, dmid, pre=fout)
        return (dx,) + grads1 + (None,) * 6 + grads2 + (None,) * 6 + (None,)


# ------------------------------------------------------------------------------------------ stem / heads
class StemFn(torch.autograd.Function):
    """Conv1d(1, 64, 7, padding=3) -- py/main16.py:134 / :177."""

    @staticmethod
    def forward(ctx, s, w, b, grad_rows=None):
        s = _frames(s, "clip batch", 1)
        B, _, T = s.shape
        y = _f32(B, 64, T, device=s.device)
        lib.wm_stem_fwd(_p(s), _p(w), _p(b), _p(y), B, T, _stream())
        ctx.save_for_backward(s, w)
        ctx.grad_rows = B if grad_rows is None else max(0, min(int(grad_rows), B))
        return y

    @staticmethod
    def backward(ctx, g):
        s, w = ctx.saved_tensors
        g = g.contiguous()
        B, _, T = s.shape
        ds = None
        if ctx.needs_input_grad[0]:        # rows >= grad_rows are known not to need a gradient (the clean half): left zero
            ds = torch.empty_like(s) if ctx.grad_rows == B else torch.zeros_like(s)
        part = _f32(2 * NCU * 512, device=s.device)
        dw, db = torch.empty_like(w), _f32(64, device=s.device)
        lib.wm_stem_bwd(_p(g), _p(s), _p(w), _p(ds), _p(part), _p(dw), _p(db), B, T, ctx.grad_rows, 0, _stream())
        return ds, dw, db, None


class Head1Fn(torch.autograd.Function):
    """Conv1d(64, 1, 1) -- Generator.decoder[2], py/main16.py:146."""

    @staticmethod
    def forward(ctx, x, w, b):
        x = _frames(x, "head input", 64)
        B, _, T = x.shape
        y = _f32(B, 1, T, device=x.device)
        lib.wm_head1_fwd(_p(x), _p(w), _p(b), _p(y), B, T, _stream())
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = g.contiguous()
        B, _, T = x.shape
        dx = torch.empty_like(x)
        part = _f32(1024 * 65, device=x.device)
        dw, db = torch.empty_like(w), _f32(1, device=x.device)
        lib.wm_head1_bwd(_p(g), _p(x), _p(w), _p(dx), _p(part), _p(dw), _p(db), B, T, 0, _stream())
        return dx, dw, db


class HeadNFn(torch.autograd.Function):
    """Conv1d(64, 1+bits, 1) followed by permute(0,2,1) -- Detector, py/main16.py:180,186.
    Returns a contiguous (B, T, 1+bits) tensor (the reference returns a view of the same shape)."""

    @staticmethod
    def forward(ctx, x, w, b):
        x = _frames(x, "head input", 64)
        B, _, T = x.shape
        NO = w.shape[0]
        if NO not in (1, 17):
            raise ValueError(f"Detector head: 1+message_bits must be 1 or 17, got {NO}")
        y = _f32(B, T, NO, device=x.device)
        lib.wm_headN_fwd(_p(x), _p(w), _p(b), _p(y), B, T, NO, _stream())
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = g.contiguous()
        B, _, T = x.shape
        NO = w.shape[0]
        dx = torch.empty_like(x)
        part = _f32(NCU * (NO * 64 + NO), device=x.device)
        dw, db = torch.empty_like(w), _f32(NO, device=x.device)
        lib.wm_headN_bwd(_p(g), _p(x), _p(w), _p(dx), _p(part), _p(dw), _p(db), B, T, NO, 0, _stream())
        return dx, dw, db


# ------------------------------------------------------------------------------------------ LSTM
_LSTM_FUSED = _os.environ.get("WM_LSTM_FUSED", "1") == "1"     # 0: separate wm_lstm_xproj + wm_lstm_fwd launches
_LSTM_BWD_FUSED = _os.environ.get("WM_LSTM_BWD_FUSED", "0") == "1"   # 1: wm_lstm_bwd + wm_lstm_dx as one launch
_LSTM = {"bwd_ws": _os.environ.get("WM_LSTM_BWD_WS", "1") == "1", "fwd_ws": True}


def set_lstm_fwd_wave_specialised(on: bool):
    """LSTM forward (wm_lstm_fwd_fused): 1 (default) the input projection of the next 32-step chunk runs on four helper waves beside the
    recurrence, 0 inside the recurrence's own waves.  Bit-identical results.  WM_LSTM_FWD_WS=0/1 sets the default."""
    lib.wm_set_lstm_fwd_wave_specialised(1 if on else 0, None)
    _LSTM["fwd_ws"] = bool(on)


if "WM_LSTM_FWD_WS" in _os.environ:
    set_lstm_fwd_wave_specialised(_os.environ["WM_LSTM_FWD_WS"] == "1")


def set_lstm_bwd_wave_specialised(on: bool):
    """LSTM backward: 1 (default) wm_lstm_bwd_wgrad -- the weight gradients are formed by four helper waves beside the recurrence, from
    the chunk of da it has just finished (T % 32 == 0, T >= 64); 0 wm_lstm_bwd followed by wm_lstm_wgrad.  WM_LSTM_BWD_WS=0/1."""
    _LSTM["bwd_ws"] = bool(on)


class LSTMFn(GradAwareFunction):
    """nn.LSTM(64,64,batch_first=True) on channel-first frames: (B,64,T) -> (B,64,T); the two permutes of
    py/main16.py:152,154 are folded into the kernels' addressing."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh):
        x = _frames(x, "LSTM input", 64)
        B, _, T = x.shape
        dev, st = x.device, _stream()
        need_grad = wants_grad(ctx)
        h = torch.empty_like(x)
        release_deferred_wgrads()                  # side stream: work queued for "beside the recurrence" starts now (eval_forward)
        if _LSTM_FUSED and T >= 8:                 # input projection inside the recurrence kernel (no xp tensor)
            gates = cst = None
            if need_grad:
                gates, cst = _f32(B, T, 256, device=dev), _f32(B, T, 64, device=dev)
            lib.wm_lstm_fwd_fused(_p(x), _p(w_ih), _p(b_ih), _p(b_hh), _p(w_hh), _p(h), _p(gates), _p(cst), B, T, st)
            if need_grad:
                ctx.gdst = _gdst(w_ih, w_hh, b_ih, b_hh)
                ctx.save_for_backward(x, h, gates, cst, w_ih, w_hh)
            return h
        xp = _f32(B, T, 256, device=dev)
        lib.wm_lstm_xproj(_p(x), _p(w_ih), _p(b_ih), _p(b_hh), _p(xp), B, T, st)
        if need_grad:
            gates, cst = xp, _f32(B, T, 64, device=dev)     # activations overwrite the projections in place
            lib.wm_lstm_fwd(_p(xp), _p(w_hh), _p(h), _p(gates), _p(cst), B, T, st)
            ctx.gdst = _gdst(w_ih, w_hh, b_ih, b_hh)
            ctx.save_for_backward(x, h, gates, cst, w_ih, w_hh)
        else:
            lib.wm_lstm_fwd(_p(xp), _p(w_hh), _p(h), None, None, B, T, st)
        return h

    @staticmethod
    def backward(ctx, dh):
        x, h, gates, cst, w_ih, w_hh = ctx.saved_tensors
        _single_backward(ctx, "LSTMFn")
        dh = dh.contiguous()
        B, _, T = x.shape
        dev, st = x.device, _stream()
        dx = torch.empty_like(x)
        release_deferred_wgrads()                  # side stream: the queued weight-gradient GEMMs run beside the recurrence
        if _LSTM["bwd_ws"] and not _LSTM_BWD_FUSED and T % 32 == 0 and T >= 64:
            part = _f32(B * (256 * 128 + 256), device=dev)
            if all(g is not None for g in ctx.gdst):           # accumulate into the flat gradient store
                gwi, gwh, gbi, gbh = ctx.gdst
                lib.wm_lstm_bwd_wgrad(_p(gates), _p(cst), _p(dh), _p(w_hh), _p(x), _p(h), _p(part), _p(gwi), _p(gwh), _p(gbi), _p(gbh),
                                      B, T, 1, st)
                lib.wm_lstm_dx(_p(gates), _p(w_ih), _p(dx), B, T, st)
                return dx, None, None, None, None
            dwi, dwh = torch.empty_like(w_ih), torch.empty_like(w_hh)
            dbi, dbh = _f32(256, device=dev), _f32(256, device=dev)
            lib.wm_lstm_bwd_wgrad(_p(gates), _p(cst), _p(dh), _p(w_hh), _p(x), _p(h), _p(part), _p(dwi), _p(dwh), _p(dbi), _p(dbh),
                                  B, T, 0, st)
            lib.wm_lstm_dx(_p(gates), _p(w_ih), _p(dx), B, T, st)
            return dx, dwi, dwh, dbi, dbh
        if _LSTM_BWD_FUSED:                        # measured: no faster than the two launches; off by default
            lib.wm_lstm_bwd_fused(_p(gates), _p(cst), _p(dh), _p(w_hh), _p(w_ih), _p(dx), B, T, st)   # gates now holds da
        else:
            lib.wm_lstm_bwd(_p(gates), _p(cst), _p(dh), _p(w_hh), B, T, st)      # gates now holds da
            lib.wm_lstm_dx(_p(gates), _p(w_ih), _p(dx), B, T, st)
        side = all(g is not None for g in ctx.gdst)
        if side:
            gwi, gwh, gbi, gbh = ctx.gdst

            def wg():
                part = _f32(NCU * (256 * 128 + 256), device=dev)
                lib.wm_lstm_wgrad(_p(gates), _p(x), _p(h), _p(part), _p(gwi), _p(gwh), _p(gbi), _p(gbh), B, T, 1, _stream())
            _on_side((gates, x, h), wg)
            return dx, None, None, None, None
        part = _f32(NCU * (256 * 128 + 256), device=dev)
        dwi, dwh = torch.empty_like(w_ih), torch.empty_like(w_hh)
        dbi, dbh = _f32(256, device=dev), _f32(256, device=dev)
        lib.wm_lstm_wgrad(_p(gates), _p(x), _p(h), _p(part), _p(dwi), _p(dwh), _p(dbi), _p(dbh), B, T, 0, st)
        return dx, dwi, dwh, dbi, dbh


# ------------------------------------------------------------------------------------------ embedding + convT
# The reference's nn.Embedding raises IndexError for an out-of-range message id (on its CUDA device: a device-side assert that
# surfaces at a later synchronisation).  Three modes (set_index_check / WM_CHECK_INDEX):
#   "sync"     (default) read the 4-byte error flag back at once and raise -- one device->host sync per Generator call;
#   "deferred" copy the flag to pinned memory asynchronously and raise at the NEXT lookup (or check_message_ids()): no
#              sync, the launch queue never drains -- what train_step uses (a mid-step sync costs the config-5 step 6 ms: its
#              LSTM chain is 200 short launches the host can only cover when it runs ahead);
#   "off"      no check (out-of-range ids read as a zero row).
_CHECK_INDEX = {"mode": {"1": "sync", "0": "off"}.get(_os.environ.get("WM_CHECK_INDEX", "1"), _os.environ.get("WM_CHECK_INDEX", "sync")),
                "pending": []}


def set_index_check(mode):
    """True / "sync" | "deferred" | False / "off" """
    _CHECK_INDEX["mode"] = {True: "sync", False: "off"}.get(mode, mode)
    if _CHECK_INDEX["mode"] not in ("sync", "deferred", "off"):
        raise ValueError("index check mode must be 'sync', 'deferred' or 'off'")


class index_check_mode:
    """context manager: run a block under another index-check mode"""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.prev = _CHECK_INDEX["mode"]
        set_index_check(self.mode)

    def __exit__(self, *exc):
        _CHECK_INDEX["mode"] = self.prev
        return False


def check_message_ids(wait=True, what="an earlier Generator call"):
    """raise IndexError if a deferred message-id check has failed (wait=False: only look at flags that have already landed)"""
    keep = []
    for ev, host, nrows in _CHECK_INDEX["pending"]:
        if not wait and not ev.query():
            keep.append((ev, host, nrows))
            continue
        ev.synchronize()
        if int(host[0]) != 0:
            _CHECK_INDEX["pending"] = []
            raise IndexError(f"message id out of range for an embedding table of {nrows} rows (deferred check of {what})")
    _CHECK_INDEX["pending"] = keep


def drop_pending_message_checks():
    """forget deferred checks that have not been read yet (their step was abandoned)"""
    _CHECK_INDEX["pending"] = []


def _note_index_error(err, nrows):
    """err: int32 device tensor (non-zero = some id was out of range), handled according to the current mode"""
    mode = _CHECK_INDEX["mode"]
    if mode == "off":
        return
    if mode == "sync":
        if int(err.item()) != 0:
            raise IndexError(f"message id out of range for an embedding table of {nrows} rows")
        return
    check_message_ids(wait=False)                    # flags of earlier calls that have landed by now
    host = torch.empty(1, dtype=torch.int32, pin_memory=True)
    host.copy_(err, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    _CHECK_INDEX["pending"].append((ev, host, nrows))


class EmbedFn(torch.autograd.Function):
    """nn.Embedding lookup, py/main16.py:158 (dense gradient like the reference's sparse=False table)."""

    @staticmethod
    def forward(ctx, table, message):
        table = _chk(table, "embedding.weight", 2)
        message = _chk(message, "message", 1, torch.int64)
        B = message.shape[0]
        vec = _f32(B, 64, device=table.device)
        err = torch.zeros(1, dtype=torch.int32, device=table.device)
        lib.wm_embed_gather(_p(table), _p(message), _p(vec), B, table.shape[0], _p(err), _stream())
        _note_index_error(err, table.shape[0])                 # nn.Embedding raises IndexError (py/main16.py:158)
        ctx.save_for_backward(message)
        ctx.nrows = table.shape[0]
        return vec

    @staticmethod
    def backward(ctx, dvec):
        (message,) = ctx.saved_tensors
        dtable = torch.zeros(ctx.nrows, 64, dtype=torch.float32, device=dvec.device)
        lib.wm_embed_scatter_add(_p(dtable), _p(message), _p(dvec.contiguous()), message.shape[0], ctx.nrows, _stream())
        return dtable, None


class ConvT7Fn(torch.autograd.Function):
    """ConvTranspose1d(64,64,7,padding=3) applied to x + emb[:, :, None]  (py/main16.py:144,156-161);
    `vec` (B,64) is the looked-up embedding row or None."""

    @staticmethod
    def forward(ctx, x, vec, w, b):
        x = _frames(x, "decoder input", 64)
        B, _, T = x.shape
        y = torch.empty_like(x)
        pro = 2 if vec is not None else 0
        if _CONV["bf16x6"]:
            if _CONV["conv7_f16x3"] and T % 128 == 0:
                lib.wm_conv64_bf7(_p(x), _p(pack_w64_h7(w, 2)), _p(vec), _p(b), _p(y), B, T, pro, 0, 1, None, _stream())
            else:
                lib.wm_conv64_bf7(_p(x), _p(pack_w64_bf7(w, 2)), _p(vec), _p(b), _p(y), B, T, pro, 0, 0, None, _stream())
        else:
            lib.wm_conv64(_p(x), None, _p(pack_w64(w, 7, 2)), _p(vec), None, None, _p(b), None, None, None, _p(y), None, B, T, 7, pro, 0,
                          _stream())
        ctx.has_vec = vec is not None
        ctx.gdst = _gdst(w, b)
        ctx.save_for_backward(x, w, vec if vec is not None else x.new_empty(0))
        return y

    @staticmethod
    def backward(ctx, g):
        x, w, vec = ctx.saved_tensors
        vec = vec if ctx.has_vec else None
        g = g.contiguous()
        B, _, T = x.shape
        dev, st = x.device, _stream()
        dx = torch.empty_like(x)
        h7 = _CONV["bf16x6"] and _CONV["conv7_f16x3"] and T % 128 == 0
        gsc = gscale_of(g) if h7 else None              # the scale both f16-split launches apply to g (.contiguous() above returns g itself)
        if h7:
            lib.wm_conv64_bf7(_p(g), _p(pack_w64_h7(w, 3)), None, None, _p(dx), B, T, 0, 3, 1, _p(gsc), st)
        elif _CONV["bf16x6"]:
            lib.wm_conv64_bf7(_p(g), _p(pack_w64_bf7(w, 3)), None, None, _p(dx), B, T, 0, 3, 0, None, st)
        else:
            lib.wm_conv64(_p(g), None, _p(pack_w64(w, 7, 3)), None, None, None, None, None, None, None, _p(dx), None, B, T, 7, 0, 3, st)
        gw, gbias = ctx.gdst
        side = gw is not None and gbias is not None

        def wg():
            part = _f32(2 * NCU * (7 * 4096 + 64), device=dev)
            if _CONV["bf16x6"]:
                lib.wm_wgrad64_bf7(_p(g), _p(x), _p(vec), _p(part), _p(gw if side else dw), _p(gbias if side else db), B, T,
                                   2 if vec is not None else 0, 1 if side else 0, 1 if h7 else 0, _p(gsc), _stream())
                return
            lib.wm_wgrad64(_p(g), None, None, None, None, _p(x), _p(vec), None, _p(part), _p(gw if side else dw),
                           _p(gbias if side else db), B, T, 7, 0, 2 if vec is not None else 0, 1, 1 if side else 0, _stream())
        dw, db = (None, None) if side else (torch.empty_like(w), _f32(64, device=dev))
        if side:
            _on_side((g, x, vec, gsc), wg)
        else:
            wg()
        dvec = None
        if vec is not None and ctx.needs_input_grad[1]:
            dvec = _f32(B, 64, device=dev)
            lib.wm_rowsum(_p(dx), _p(dvec), B * 64, T, st)
        return dx, dvec, dw, db


# ------------------------------------------------------------------------------------------ delta post-processing
class PostprocFn(torch.autograd.Function):
    """stages bit0 fir_lowpass | bit1 clamp_peak | bit2 limit_rms, fused (py/main16.py:53-72, :245-247)."""

    @staticmethod
    def forward(ctx, delta, taps, thr, max_rms, eps, stages):
        delta = _frames(delta, "delta", 1)
        B, _, T = delta.shape
        dev = delta.device
        out = torch.empty_like(delta)
        need = ctx.needs_input_grad[0]
        f = torch.empty_like(delta) if need else None
        stats = _f32(B, 2, device=dev)
        lib.wm_postproc_fwd(_p(delta), _p(taps), taps.numel(), thr, max_rms, eps, stages, _p(f), _p(out), _p(stats), B, T, _stream())
        if need:
            ctx.save_for_backward(f, stats, taps)
        ctx.cfg = (thr, max_rms, stages)
        return out

    @staticmethod
    def backward(ctx, g):
        f, stats, taps = ctx.saved_tensors
        thr, max_rms, stages = ctx.cfg
        g = g.contiguous()
        B, _, T = f.shape
        d = torch.empty_like(f)
        lib.wm_postproc_bwd(_p(g), _p(f), _p(stats), _p(taps), taps.numel(), thr, max_rms, stages, _p(d), B, T, _stream())
        return d, None, None, None, None, None


# ------------------------------------------------------------------------------------------ losses
class _SpectralLossFn(torch.autograd.Function):
    """Shared shape of the three STFT losses: the kernel returns the loss AND d loss / d signal, the tape
    only scales the cached gradient by the incoming scalar."""

    @staticmethod
    def forward(ctx, kind, clean, sig, tables):
        sig2 = _chk(sig, "signal", 3)
        B, _, T = sig2.shape
        dev, st = sig2.device, _stream()
        need = ctx.needs_input_grad[2]
        n_fft, hop = {"mel": (1024, 256), "loud": (2048, 512), "hf": (512, 128)}[kind]
        F = 1 + T // hop
        loss = _f32(1, device=dev)
        part = _f32(B * F, device=dev)
        gfr = _f32(B * F * n_fft, device=dev) if need else None
        dsig = torch.empty_like(sig2) if need else None
        if kind == "mel":
            c2 = _chk(clean, "clean", 3)
            fb, klo, khi, mlo = tables
            lib.wm_mel_loss(_p(c2), _p(sig2), _p(fb), _p(klo), _p(khi), _p(mlo), _p(gfr), _p(part), _p(loss), _p(dsig), B, T, st)
        elif kind == "loud":
            c2 = _chk(clean, "clean", 3)
            lib.wm_loud_loss(_p(c2), _p(sig2), 0.01, _p(gfr), _p(part), _p(loss), _p(dsig), B, T, st)
        else:
            lib.wm_hf_penalty(_p(sig2), int(tables), _p(gfr), _p(part), _p(loss), _p(dsig), B, T, st)
        if need:
            ctx.save_for_backward(dsig)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dsig,) = ctx.saved_tensors
        return None, None, dsig * g, None


class BCEFn(torch.autograd.Function):
    """loc_loss and bce of py/main16.py:252-264 from one pass over the (2B,T,1+bits) logits."""

    @staticmethod
    def forward(ctx, logits, message):
        logits = _chk(logits, "logits", 3)
        message = _chk(message, "message", 1, torch.int64)
        R, T, NO = logits.shape
        B = message.shape[0]
        if R != 2 * B:
            raise ValueError(f"logits must hold [watermarked; clean] = 2*B clips, got {R} for B={B}")
        dev = logits.device
        part = _f32(2 * R * ((T * NO + 4095) // 4096), device=dev)
        out = torch.zeros(2, dtype=torch.float32, device=dev)
        lib.wm_bce_fwd(_p(logits), _p(message), _p(part), _p(out[0]), _p(out[1]), B, R, T, NO, _stream())
        ctx.save_for_backward(logits, message)
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_loc, g_bce):
        logits, message = ctx.saved_tensors
        R, T, NO = logits.shape
        gl = g_loc.contiguous().float().reshape(1) if g_loc is not None else torch.zeros(1, device=logits.device)
        gb = g_bce.contiguous().float().reshape(1) if g_bce is not None else torch.zeros(1, device=logits.device)
        d = torch.empty_like(logits)
        lib.wm_bce_bwd(_p(logits), _p(message), _p(gl), _p(gb), _p(d), message.shape[0], R, T, NO, _stream())
        return d, None


class L1Fn(torch.autograd.Function):
    """F.l1_loss(delta, 0), py/main16.py:266."""

    @staticmethod
    def forward(ctx, x):
        x = _chk(x, "delta")
        out = _f32(1, device=x.device)
        part = _f32(256, device=x.device)
        lib.wm_l1_fwd(_p(x), _p(part), _p(out), x.numel(), _stream())
        ctx.save_for_backward(x)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        lib.wm_l1_bwd(_p(x), _p(g.contiguous().float().reshape(1)), _p(dx), x.numel(), _stream())
        return dx
