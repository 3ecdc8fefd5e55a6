"""diagnostic: where a conv64 workgroup spends its cycles (needs the -DWM_STAMP build libwm_hip_stamp.so)"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = os.environ.get("WM_STAMP_LIB") or os.path.join(ROOT, "audio-watermarking-deep-learning-watermarks-for-authenticating-speech_amd", "libwm_hip_stamp.so")
L = ctypes.CDLL(so)
dev = torch.device("cuda:0"); B, T = 256, 16000
x = torch.randn(B, 64, T, device=dev); y = torch.empty_like(x); x2 = torch.randn(B, 64, T, device=dev); x3 = torch.randn(B, 64, T, device=dev)
w = torch.randn(64, 64, 3, device=dev) * 0.05; wp = torch.empty(3 * 4096, device=dev); bias = torch.randn(64, device=dev)
c = [torch.rand(128, device=dev) for _ in range(5)]
stats = torch.empty(256 * 128, device=dev)
buf = torch.zeros(256 * 4 * 6, dtype=torch.int64, device=dev)
vp = ctypes.c_void_p
L.wm_debug_set_stamp_buffer(vp(buf.data_ptr()))
L.wm_pack_w64(vp(w.data_ptr()), vp(wp.data_ptr()), 3, 0, None)
wpb = torch.empty(3 * 3 * 4096, dtype=torch.int16, device=dev)
L.wm_pack_w64_bf(vp(w.data_ptr()), vp(wpb.data_ptr()), 0, None)
ARITH = int(os.environ.get("WM_DIAG_ARITH", "0"))      # 1: the f16 two-piece build of the forward variants
wph = torch.empty(2 * 3 * 4096 + 4, dtype=torch.int16, device=dev)
L.wm_pack_w64_h(vp(w.data_ptr()), vp(wph.data_ptr()), 0, None)
def HARITH(pro, epi):
    return ARITH if pro in (0, 1) and epi == 0 else 0
def run_bf(name, pro, epi, st):
    args = [vp(x.data_ptr()), vp(x2.data_ptr()) if pro == 3 else None, vp((wph if HARITH(pro, epi) else wpb).data_ptr()),
            vp(c[0].data_ptr()), vp(c[1].data_ptr()), vp(c[2].data_ptr()), vp(bias.data_ptr()),
            vp(x3.data_ptr()) if epi in (1, 2) else None, vp(c[3].data_ptr()), vp(c[4].data_ptr()), vp(y.data_ptr()),
            vp(stats.data_ptr()) if st else None, B, T, pro, epi, HARITH(pro, epi), None]
    for _ in range(2):
        buf.zero_(); rc = L.wm_conv64_bf(*args); torch.cuda.synchronize()
    assert rc == 0, rc
    d = buf.view(256, 4, 6).double().mean(dim=(0, 1))
    names = ["load-issue+e1", "mfma", "epilogue", "bar1", "lds-write", "bar2"]
    print(f"bf {name:25s} " + "  ".join(f"{n} {v:9.0f}" for n, v in zip(names, d)) + f"   total {d.sum():9.0f}")
def run(name, pro, epi, st):
    args = [vp(x.data_ptr()), vp(x2.data_ptr()) if pro == 3 else None, vp(wp.data_ptr()),
            vp(c[0].data_ptr()), vp(c[1].data_ptr()), vp(c[2].data_ptr()), vp(bias.data_ptr()),
            vp(x3.data_ptr()) if epi in (1, 2) else None, vp(c[3].data_ptr()), vp(c[4].data_ptr()), vp(y.data_ptr()),
            vp(stats.data_ptr()) if st else None, B, T, 3, pro, epi, None]
    for _ in range(2):
        buf.zero_(); rc = L.wm_conv64(*args); torch.cuda.synchronize()
    assert rc == 0, rc
    d = buf.view(256, 4, 6).double().mean(dim=(0, 1))
    names = ["load-issue+e1", "mfma", "epilogue", "bar1", "lds-write", "bar2"]
    print(f"{name:28s} " + "  ".join(f"{n} {v:9.0f}" for n, v in zip(names, d)) + f"   total {d.sum():9.0f}")
if len(sys.argv) > 1:
    L.wm_set_conv_bf_schedule(2, None)
    def run_bf3(name, pro, epi, st):
        args = [vp(x.data_ptr()), vp(x2.data_ptr()) if pro == 3 else None, vp((wph if HARITH(pro, epi) else wpb).data_ptr()),
                vp(c[0].data_ptr()), vp(c[1].data_ptr()), vp(c[2].data_ptr()), vp(bias.data_ptr()),
                vp(x3.data_ptr()) if epi in (1, 2) else None, vp(c[3].data_ptr()), vp(c[4].data_ptr()), vp(y.data_ptr()),
                vp(stats.data_ptr()) if st else None, B, T, pro, epi, HARITH(pro, epi), None]
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        for _ in range(2):
            buf.zero_(); torch.cuda.synchronize(); e0.record(); rc = L.wm_conv64_bf(*args); e1.record(); torch.cuda.synchronize()
        assert rc == 0, rc
        ms = e0.elapsed_time(e1)
        d = buf.view(256, 4, 6).double().mean(dim=(0, 1)) / 125
        names = ["to h=3", "to h=15", "mfma loop", "epilogue", "barrier", "-"]
        tot = float(d[2] + d[3] + d[4]) * 125
        print(f"bf3 {name:25s} " + "  ".join(f"{n} {v:9.0f}" for n, v in zip(names, d)) + f"  | {ms:.3f} ms, {tot/ms/1e6:.2f} GHz effective")
    run_bf3("fwd none/bias", 0, 0, False)
    run_bf3("fwd none/bias +stats", 0, 0, True)
    run_bf3("fwd bnrelu/bias +stats", 1, 0, True)
    run_bf3("dgrad bnbwd/relumask +stats", 3, 1, True)
    run_bf3("dgrad bnbwd/add", 3, 2, False)
    sys.exit(0)
run("fwd none/bias", 0, 0, False)
run("fwd none/bias +stats", 0, 0, True)
run("fwd bnrelu/bias +stats", 1, 0, True)
run("dgrad bnbwd/relumask +stats", 3, 1, True)
run("dgrad bnbwd/add", 3, 2, False)
run("dgrad bnbwd/none", 3, 3, False)
run_bf("fwd none/bias", 0, 0, False)
run_bf("fwd none/bias +stats", 0, 0, True)
run_bf("fwd bnrelu/bias +stats", 1, 0, True)
run_bf("dgrad bnbwd/relumask +stats", 3, 1, True)
run_bf("dgrad bnbwd/add", 3, 2, False)
