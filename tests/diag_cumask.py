"""Diagnostic (not a test): do CU-masked HIP streams let the LSTM recurrence and the weight-gradient GEMMs share the chip?

The BPTT recurrence holds one 4-wave workgroup per clip (256 clips -> one per CU) and is latency bound; weight-gradient
GEMMs launched beside it on an ordinary stream slow it down by as much as they gain.  Here each
side gets its own CUs: hipExtStreamCreateWithCUMask.

run:  python tests/diag_cumask.py
"""
import ctypes
import sys
import os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import awm_amd  # noqa: E402
from awm_amd import ops  # noqa: E402
from awm_amd._lib import lib  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(words):
    s = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(len(words)), arr)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {rc}")
    return torch.cuda.ExternalStream(s.value)


def p(t):
    return None if t is None else t.data_ptr()


def main():
    dev = torch.device("cuda:0")
    B, T = 256, 16000
    torch.manual_seed(0)
    gates0 = torch.rand(B, T, 256, device=dev) * 0.8 + 0.1
    cst = torch.randn(B, T, 64, device=dev) * 0.5
    dh = torch.randn(B, 64, T, device=dev) * 0.01
    w_hh = torch.randn(256, 64, device=dev) * 0.1
    gates = gates0.clone()

    # weight-gradient operands of a Detector-side ResBlock (2B clips)
    B2 = 2 * B
    dz = torch.randn(B2, 64, T, device=dev)
    y = torch.randn(B2, 64, T, device=dev)
    y1 = torch.randn(B2, 64, T, device=dev)
    k = torch.randn(4, 64, device=dev) * 0.1
    sc = torch.rand(64, device=dev) + 0.5
    sh = torch.randn(64, device=dev) * 0.1
    wpart = torch.empty(2 * 256 * (3 * 4096 + 64), device=dev)
    dw = torch.zeros(64, 64, 3, device=dev)
    db = torch.zeros(64, device=dev)

    def lstm():
        lib.wm_lstm_bwd(p(gates), p(cst), p(dh), p(w_hh), B, T, torch.cuda.current_stream().cuda_stream)

    def wgrad(n=1):
        for _ in range(n):
            lib.wm_wgrad64_bf(p(dz), p(y), p(k[0]), p(k[1]), p(k[3]), p(y1), p(sc), p(sh), p(wpart), p(dw), p(db), B2, T, 3, 1, 0,
                              torch.cuda.current_stream().cuda_stream)

    def timed(fn, stream, reps=3):
        best = 1e9
        for _ in range(reps):
            gates.copy_(gates0)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(stream):
                e0.record()
                fn()
                e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return best

    def both(s_l, s_w, nw, reps=3):
        best = (1e9, 0, 0)
        for _ in range(reps):
            gates.copy_(gates0)
            torch.cuda.synchronize()
            main_s = torch.cuda.current_stream()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            l0, l1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(main_s)
            s_l.wait_event(e0)
            s_w.wait_event(e0)
            with torch.cuda.stream(s_l):
                l0.record()
                lstm()
                l1.record()
            with torch.cuda.stream(s_w):
                w0.record()
                wgrad(nw)
                w1.record()
            main_s.wait_event(l1)
            main_s.wait_event(w1)
            e1.record(main_s)
            torch.cuda.synchronize()
            tot = e0.elapsed_time(e1)
            if tot < best[0]:
                best = (tot, l0.elapsed_time(l1), w0.elapsed_time(w1))
        return best

    full = torch.cuda.Stream()
    full2 = torch.cuda.Stream()
    F = 0xffffffff
    masks = {
        "first128": ([F, F, F, F, 0, 0, 0, 0], [0, 0, 0, 0, F, F, F, F]),
        "even": ([0x55555555] * 8, [0xaaaaaaaa] * 8),
        "first64": ([F, F, 0, 0, 0, 0, 0, 0], [0, 0, F, F, F, F, F, F]),
        "first96": ([F, F, F, 0, 0, 0, 0, 0], [0, 0, 0, F, F, F, F, F]),
        "first192": ([F, F, F, F, F, F, 0, 0], [0, 0, 0, 0, 0, 0, F, F]),
    }
    print(f"alone, unmasked: lstm_bwd {timed(lstm, full):.3f} ms   wgrad x1 {timed(lambda: wgrad(1), full):.3f} ms   "
          f"wgrad x4 {timed(lambda: wgrad(4), full):.3f} ms", flush=True)
    t = both(full, full2, 4)
    print(f"together, unmasked streams: total {t[0]:.3f} ms (lstm {t[1]:.3f}, wgrad x4 {t[2]:.3f})", flush=True)
    for name, (ml, mw) in masks.items():
        sl, sw = masked_stream(ml), masked_stream(mw)
        a = timed(lstm, sl)
        b = timed(lambda: wgrad(4), sw)
        t = both(sl, sw, 4)
        print(f"mask {name:9s}: lstm alone on its CUs {a:.3f} ms, wgrad x4 alone on the rest {b:.3f} ms, together total {t[0]:.3f} ms "
              f"(lstm {t[1]:.3f}, wgrad x4 {t[2]:.3f})", flush=True)
        t = both(sl, full2, 4)
        print(f"      {name:9s}: lstm masked + wgrad unmasked: total {t[0]:.3f} ms (lstm {t[1]:.3f}, wgrad x4 {t[2]:.3f})", flush=True)


if __name__ == "__main__":
    main()
