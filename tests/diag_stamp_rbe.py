"""diagnostic: where a resblock_eval workgroup spends its cycles (needs the -DWM_STAMP build of csrc/conv64.hip:
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DWM_STAMP -shared csrc/conv64.hip -o libwm_hip_stamp.so)"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = os.environ.get("WM_STAMP_LIB") or os.path.join(ROOT, "audio-watermarking-deep-learning-watermarks-for-authenticating-speech_amd", "libwm_hip_stamp.so")
L = ctypes.CDLL(so)
dev = torch.device("cuda:0"); B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 16000
x = torch.randn(B, 64, T, device=dev).abs(); y = torch.empty_like(x)
w1 = torch.randn(64, 64, 3, device=dev) * 0.05; w2 = torch.randn(64, 64, 3, device=dev) * 0.05
c = [torch.rand(64, device=dev) for _ in range(6)]
wm_scaled = True
buf = torch.zeros(256 * 4 * 6, dtype=torch.int64, device=dev)
vp = ctypes.c_void_p
L.wm_debug_set_stamp_buffer(vp(buf.data_ptr()))
wp1 = torch.empty(3 * 3 * 4096, dtype=torch.int16, device=dev); wp2 = torch.empty_like(wp1)
L.wm_pack_w64_bf_scaled(vp(w1.data_ptr()), vp(c[1].data_ptr()), vp(wp1.data_ptr()), None); L.wm_pack_w64_bf_scaled(vp(w2.data_ptr()), vp(c[4].data_ptr()), vp(wp2.data_ptr()), None)
args = [vp(x.data_ptr()), vp(wp1.data_ptr()), vp(wp2.data_ptr())] + [vp(t.data_ptr()) for t in c] + [vp(y.data_ptr()), B, T, 0, None]
e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
for _ in range(3):
    buf.zero_(); torch.cuda.synchronize(); e0.record(); rc = L.wm_resblock_eval_bf(*args); e1.record(); torch.cuda.synchronize()
assert rc == 0, rc
ms = e0.elapsed_time(e1)
ntile = B * ((T + 2 + 123) // 124) / 256
d = buf.view(256, 4, 6).double().mean(dim=(0, 1)) / ntile
names = ["conv1 blk0", "conv1 blk1", "epi1 serial", "barrier1", "conv2", "epi2+barrier2"]
print(f"B={B}: {ms:.3f} ms, {ntile:.1f} tiles per workgroup, per tile: " + "  ".join(f"{n} {v:7.0f}" for n, v in zip(names, d)) +
      f"  total {float(d.sum()):7.0f} stamp ticks = {ms*1e3/ntile:.2f} us")
