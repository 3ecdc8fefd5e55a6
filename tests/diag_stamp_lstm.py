"""diagnostic: cycle budget of ONE step of the 16 000-step recurrences (needs the -DWM_STAMP build of csrc/lstm.hip:
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DWM_STAMP -shared csrc/lstm.hip -o libwm_lstm_stamp.so).  The stamps serialise
LDS traffic at each point (s_waitcnt lgkmcnt(0)), so the stamped step is longer than the real one; the shares are what to read."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pkg = os.path.join(ROOT, "audio-watermarking-deep-learning-watermarks-for-authenticating-speech_amd")
S = ctypes.CDLL(os.environ.get("WM_LSTM_STAMP_LIB") or os.path.join(pkg, "libwm_lstm_stamp.so"))
P = ctypes.CDLL(os.path.join(pkg, "libwm_hip.so"))
dev = torch.device("cuda:0"); B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 16000
vp = ctypes.c_void_p
g = torch.Generator().manual_seed(1)
x = (torch.randn(B, 64, T, generator=g) * 0.5).to(dev)
w_ih, w_hh = (torch.randn(256, 64, generator=g) * 0.12).to(dev), (torch.randn(256, 64, generator=g) * 0.12).to(dev)
b_ih, b_hh = (torch.randn(256, generator=g) * 0.1).to(dev), (torch.randn(256, generator=g) * 0.1).to(dev)
h = torch.empty(B, 64, T, device=dev); gates = torch.empty(B, T, 256, device=dev); cst = torch.empty(B, T, 64, device=dev)
dh = (torch.randn(B, 64, T, generator=g) * 0.1).to(dev)
buf = torch.zeros(B * 4 * 6, dtype=torch.int64, device=dev)
S.wm_debug_set_lstm_stamp_buffer(vp(buf.data_ptr()))


def timed(fn):
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    fn(); torch.cuda.synchronize(); e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)


def fwd(L):
    return L.wm_lstm_fwd_fused(vp(x.data_ptr()), vp(w_ih.data_ptr()), vp(b_ih.data_ptr()), vp(b_hh.data_ptr()), vp(w_hh.data_ptr()),
                               vp(h.data_ptr()), vp(gates.data_ptr()), vp(cst.data_ptr()), B, T, None)


def report(name, ms_plain, ms_stamp, names):
    d = buf.view(B, 4, 6).double().mean(dim=(0, 1)) / T
    tot = float(d[:len(names)].sum())
    print(f"{name}: B={B} T={T}  unstamped {ms_plain:.3f} ms = {ms_plain * 1e6 / T:.0f} ns/step; stamped {ms_stamp:.3f} ms = "
          f"{ms_stamp * 1e6 / T:.0f} ns/step, {tot:.0f} ticks/step ({tot / (ms_stamp * 1e6 / T):.2f} ticks/ns)")
    for n, v in zip(names, d):
        print(f"    {n:58s} {float(v):7.0f} ticks  {100 * float(v) / tot:5.1f} %")


ms_plain = timed(lambda: fwd(P))
buf.zero_()
ms_stamp = timed(lambda: fwd(S))
report("lstm_fwd_fused_kernel<true>", ms_plain, ms_stamp,
       ["LDS -> registers: h(t-1), projection x(t)", "recurrent dot product (64 v_fmac_f32_dpp row_newbcast + 2 MFMA)", "gate activation (exp, rcp)",
        "quad exchange, cell update, tanh(c), h", "h -> LDS, stores of gates / c / h issued", "s_waitcnt + s_barrier"])
gsave = gates.clone()


def bwd(L):
    gates.copy_(gsave)
    return L.wm_lstm_bwd(vp(gates.data_ptr()), vp(cst.data_ptr()), vp(dh.data_ptr()), vp(w_hh.data_ptr()), B, T, None)


def timed_bwd(L):
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    bwd(L); torch.cuda.synchronize(); gates.copy_(gsave); torch.cuda.synchronize()
    e0.record(); L.wm_lstm_bwd(vp(gates.data_ptr()), vp(cst.data_ptr()), vp(dh.data_ptr()), vp(w_hh.data_ptr()), B, T, None); e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


ms_plain = timed_bwd(P)
buf.zero_()
ms_stamp = timed_bwd(S)
report("lstm_bwd_kernel", ms_plain, ms_stamp,
       ["LDS -> registers: four partial dh (+ the dh-independent gate factors), summed", "dc, da", "W_hh^T da (3 half / row swaps + 64 v_fmac_f32_dpp)",
        "partial dh -> LDS", "s_waitcnt + s_barrier"])
