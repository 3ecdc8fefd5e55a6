"""diagnostic: where a dwgrad64bf workgroup spends its cycles (needs the -DWM_STAMP build of csrc/conv64.hip:
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -pragma-unroll-threshold=100000 -DWM_STAMP -shared csrc/conv64.hip -o libwm_hip_stamp.so)"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = os.environ.get("WM_STAMP_LIB") or os.path.join(ROOT, "audio-watermarking-deep-learning-watermarks-for-authenticating-speech_amd", "libwm_hip_stamp.so")
L = ctypes.CDLL(so)
dev = torch.device("cuda:0"); B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 16000
dz = torch.randn(B, 64, T, device=dev); y2 = torch.randn(B, 64, T, device=dev); y1 = torch.randn(B, 64, T, device=dev)
if len(sys.argv) > 2 and sys.argv[2] == "zeros":        # same instruction stream on all-zero operands: the clock the chip then holds
    dz.zero_(); y2.zero_(); y1.zero_()                  # separates power-limited from issue-limited time
if len(sys.argv) > 2 and sys.argv[2] == "sparse":       # ReLU-like operands (half of the activations zero)
    y1.clamp_(min=0); dz.mul_((torch.rand_like(dz) > 0.5).float())
out = torch.empty_like(dz)
w = torch.randn(64, 64, 3, device=dev) * 0.05
k = torch.rand(4, 64, device=dev); sc = torch.rand(64, device=dev) + 0.5; sh = torch.randn(64, device=dev) * 0.1
stats = torch.empty(256 * 128, device=dev); wpart = torch.empty(256 * (3 * 4096 + 64), device=dev)
dw = torch.empty(64, 64, 3, device=dev); db = torch.empty(64, device=dev)
buf = torch.zeros(256 * 4 * 6, dtype=torch.int64, device=dev)
vp = ctypes.c_void_p
L.wm_debug_set_stamp_buffer(vp(buf.data_ptr()))
ARITH = 1 if os.environ.get("WM_DIAG_ARITH", "1") == "1" else 0       # 1: f16 two-piece split (default backward), 0: bf16x6
wpb = torch.empty(3 * 3 * 4096 + 4, dtype=torch.int16, device=dev)
gsc = torch.tensor([1.0, 1.0], device=dev)
(L.wm_pack_w64_h if ARITH else L.wm_pack_w64_bf)(vp(w.data_ptr()), vp(wpb.data_ptr()), 1, None)
def run(name, xpro, epi):
    args = [vp(dz.data_ptr()), vp(y2.data_ptr()), vp(k[0].data_ptr()), vp(k[1].data_ptr()), vp(k[3].data_ptr()), vp(wpb.data_ptr()),
            vp(y1.data_ptr()), vp(sc.data_ptr()) if xpro else None, vp(sh.data_ptr()) if xpro else None,
            vp(y1.data_ptr()), vp(sc.data_ptr()) if epi == 1 else None, vp(sh.data_ptr()) if epi == 1 else None,
            vp(out.data_ptr()), vp(stats.data_ptr()) if epi == 1 else None, vp(wpart.data_ptr()), vp(dw.data_ptr()), vp(db.data_ptr()),
            B, T, xpro, epi, 0, None, ARITH, vp(gsc.data_ptr()) if ARITH else None, None, None]
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    for _ in range(3):
        buf.zero_(); torch.cuda.synchronize(); e0.record(); rc = L.wm_dwgrad64_bf(*args); e1.record(); torch.cuda.synchronize()
    assert rc == 0, rc
    ms = e0.elapsed_time(e1)
    ntile = B * (T // 64) / 256
    d = buf.view(256, 4, 6).double().mean(dim=(0, 1)) / ntile
    names = ["phase A (dgrad)", "barrier 1", "phase B (wgrad)", "barrier 2", "(of barrier 1: LDS drain)", "(of phase B: first two k-blocks)"]
    print(f"{name}: B={B} {ms:.3f} ms, per 64-step tile: " + "  ".join(f"{n} {v:6.0f}" for n, v in zip(names, d)) +
          f"  total {float(d[:4].sum()):6.0f} ticks = {ms*1e3/ntile:.2f} us")
    pw = buf.view(256, 4, 6).double().mean(dim=0) / ntile          # per wave
    for w in range(4):
        print(f"    wave {w}: " + "  ".join(f"{n} {float(v):6.0f}" for n, v in zip(names, pw[w])))
run("conv2 pair (bnrelu / relumask+stats)", 1, 1)
run("conv1 pair (none / add)", 0, 2)
