"""micro-benchmark of individual C-ABI launches at bench size (not a test): python tests/bench_kernels.py [B]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import awm_amd
from awm_amd import ops
from awm_amd.ops import _p, _stream, _f32, lib
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = 16000
only = sys.argv[2] if len(sys.argv) > 2 else ""
torch.manual_seed(0)
x = torch.randn(B, 64, T, device=dev) * 0.5
x2 = torch.randn(B, 64, T, device=dev) * 0.5
x3 = torch.randn(B, 64, T, device=dev) * 0.5
y = torch.empty_like(x)
w3 = torch.randn(64, 64, 3, device=dev) * 0.05
w7 = torch.randn(64, 64, 7, device=dev) * 0.05
c = [torch.rand(128, device=dev) + 0.5 for _ in range(6)]
bias = torch.randn(64, device=dev)
stats = _f32(256 * 128, device=dev)
wp3 = ops.pack_w64(w3, 3, 0); wp7 = ops.pack_w64(w7, 7, 2)
vec = torch.randn(B, 64, device=dev)
wpart = _f32(512 * (7 * 4096 + 64), device=dev)
dw3 = torch.empty_like(w3); dw7 = torch.empty_like(w7); db = _f32(64, device=dev)
st = _stream()
def timeit(name, fn, flops=None, bytes_=None, n=5):
    if only and only not in name: return
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    extra = ""
    if flops: extra += f"  {flops/ms/1e9:7.1f} TFLOP/s ({flops/ms/1e9/157.3*100:4.1f}% of fp32 MFMA)"
    if bytes_: extra += f"  {bytes_/ms/1e9*1e3/1e3:6.2f} TB/s"
    print(f"{name:42s} {ms:8.3f} ms{extra}", flush=True)
F3 = 2.0 * 64 * 64 * 3 * T * B; F7 = 2.0 * 64 * 64 * 7 * T * B; FR = 4.0 * 64 * T * B
timeit("conv64 k3 fwd none/bias", lambda: lib.wm_conv64(_p(x), None, _p(wp3), None, None, None, _p(bias), None, None, None, _p(y), None, B, T, 3, 0, 0, st), F3, 2 * FR)
timeit("conv64 k3 fwd none/bias +stats", lambda: lib.wm_conv64(_p(x), None, _p(wp3), None, None, None, _p(bias), None, None, None, _p(y), _p(stats), B, T, 3, 0, 0, st), F3, 2 * FR)
timeit("conv64 k3 fwd bnrelu/bias +stats", lambda: lib.wm_conv64(_p(x), None, _p(wp3), _p(c[0]), _p(c[1]), None, _p(bias), None, None, None, _p(y), _p(stats), B, T, 3, 1, 0, st), F3, 2 * FR)
timeit("conv64 k3 dgrad bnbwd/relumask +stats", lambda: lib.wm_conv64(_p(x), _p(x2), _p(wp3), _p(c[0]), _p(c[1]), _p(c[2]), None, _p(x3), _p(c[3]), _p(c[4]), _p(y), _p(stats), B, T, 3, 3, 1, st), F3, 4 * FR)
timeit("conv64 k3 dgrad bnbwd/add", lambda: lib.wm_conv64(_p(x), _p(x2), _p(wp3), _p(c[0]), _p(c[1]), _p(c[2]), None, _p(x3), None, None, _p(y), None, B, T, 3, 3, 2, st), F3, 4 * FR)
wpb = ops.pack_w64_bf(w3, 0)
timeit("conv64bf k3 fwd none/bias", lambda: lib.wm_conv64_bf(_p(x), None, _p(wpb), None, None, None, _p(bias), None, None, None, _p(y), None, B, T, 0, 0, 0, st), F3, 2 * FR)
timeit("conv64bf k3 fwd none/bias +stats", lambda: lib.wm_conv64_bf(_p(x), None, _p(wpb), None, None, None, _p(bias), None, None, None, _p(y), _p(stats), B, T, 0, 0, 0, st), F3, 2 * FR)
timeit("conv64bf k3 fwd bnrelu/bias +stats", lambda: lib.wm_conv64_bf(_p(x), None, _p(wpb), _p(c[0]), _p(c[1]), None, _p(bias), None, None, None, _p(y), _p(stats), B, T, 1, 0, 0, st), F3, 2 * FR)
timeit("conv64bf k3 dgrad bnbwd/relumask +stats", lambda: lib.wm_conv64_bf(_p(x), _p(x2), _p(wpb), _p(c[0]), _p(c[1]), _p(c[2]), None, _p(x3), _p(c[3]), _p(c[4]), _p(y), _p(stats), B, T, 3, 1, 0, st), F3, 4 * FR)
timeit("conv64bf k3 dgrad bnbwd/add", lambda: lib.wm_conv64_bf(_p(x), _p(x2), _p(wpb), _p(c[0]), _p(c[1]), _p(c[2]), None, _p(x3), None, None, _p(y), None, B, T, 3, 2, 0, st), F3, 4 * FR)
if not only or "acc" in only:
    # accuracy of both arithmetic modes against fp64 on one clip
    xs = x[:2].double().cpu(); ref = torch.nn.functional.conv1d(xs, w3.double().cpu(), bias.double().cpu(), padding=1)
    y1_ = torch.empty_like(x[:2]); y2_ = torch.empty_like(x[:2])
    lib.wm_conv64(_p(x[:2].contiguous()), None, _p(wp3), None, None, None, _p(bias), None, None, None, _p(y1_), None, 2, T, 3, 0, 0, st)
    lib.wm_conv64_bf(_p(x[:2].contiguous()), None, _p(wpb), None, None, None, _p(bias), None, None, None, _p(y2_), None, 2, T, 0, 0, 0, st)
    cpu32 = torch.nn.functional.conv1d(x[:2].cpu(), w3.cpu(), bias.cpu(), padding=1)
    sc = float(ref.abs().max())
    print(f"acc vs fp64 (max abs / max|ref|): native fp32 MFMA {float((y1_.double().cpu()-ref).abs().max())/sc:.2e}   bf16x6 {float((y2_.double().cpu()-ref).abs().max())/sc:.2e}   torch CPU fp32 {float((cpu32.double()-ref).abs().max())/sc:.2e}")
timeit("conv64 k7 fwd addvec/bias", lambda: lib.wm_conv64(_p(x), None, _p(wp7), _p(vec), None, None, _p(bias), None, None, None, _p(y), None, B, T, 7, 2, 0, st), F7, 2 * FR)
wpb7 = ops.pack_w64_bf7(w7, 2)
timeit("conv64bf7 k7 fwd addvec/bias", lambda: lib.wm_conv64_bf7(_p(x), _p(wpb7), _p(vec), _p(bias), _p(y), B, T, 2, 0, 0, None, st), F7, 2 * FR)
timeit("conv64bf7 k7 dgrad none/none", lambda: lib.wm_conv64_bf7(_p(x), _p(wpb7), None, None, _p(y), B, T, 0, 3, 0, None, st), F7, 2 * FR)
timeit("conv64 k7 dgrad none/none", lambda: lib.wm_conv64(_p(x), None, _p(wp7), None, None, None, None, None, None, None, _p(y), None, B, T, 7, 0, 3, st), F7, 2 * FR)
timeit("wgrad64 k3 bnbwd x bnrelu", lambda: lib.wm_wgrad64(_p(x), _p(x2), _p(c[0]), _p(c[1]), _p(c[2]), _p(x3), _p(c[3]), _p(c[4]), _p(wpart), _p(dw3), _p(db), B, T, 3, 3, 1, 0, 0, st), F3, 3 * FR)
timeit("wgrad64 k3 bnbwd x none", lambda: lib.wm_wgrad64(_p(x), _p(x2), _p(c[0]), _p(c[1]), _p(c[2]), _p(x3), None, None, _p(wpart), _p(dw3), _p(db), B, T, 3, 3, 0, 0, 0, st), F3, 3 * FR)
timeit("wgrad64bf k3 bnbwd x bnrelu", lambda: lib.wm_wgrad64_bf(_p(x), _p(x2), _p(c[0]), _p(c[1]), _p(c[2]), _p(x3), _p(c[3]), _p(c[4]), _p(wpart), _p(dw3), _p(db), B, T, 3, 1, 0, st), F3, 3 * FR)
timeit("wgrad64bf k3 bnbwd x none", lambda: lib.wm_wgrad64_bf(_p(x), _p(x2), _p(c[0]), _p(c[1]), _p(c[2]), _p(x3), None, None, _p(wpart), _p(dw3), _p(db), B, T, 3, 0, 0, st), F3, 3 * FR)
timeit("wgrad64 k7 none x addvec", lambda: lib.wm_wgrad64(_p(x), None, None, None, None, _p(x3), _p(vec), None, _p(wpart), _p(dw7), _p(db), B, T, 7, 0, 2, 1, 0, st), F7, 2 * FR)
wph7 = ops.pack_w64_h7(w7, 2); gsc7 = ops.gscale_absmax(x)
timeit("conv64bf7 k7 fwd addvec/bias  f16x3", lambda: lib.wm_conv64_bf7(_p(x), _p(wph7), _p(vec), _p(bias), _p(y), B, T, 2, 0, 1, None, st), F7, 2 * FR)
timeit("conv64bf7 k7 dgrad none/none  f16x3", lambda: lib.wm_conv64_bf7(_p(x), _p(wph7), None, None, _p(y), B, T, 0, 3, 1, _p(gsc7), st), F7, 2 * FR)
timeit("gscale_absmax", lambda: ops.gscale_absmax(x), None, FR)
timeit("wgrad64bf7 k7 none x addvec  f16x3", lambda: lib.wm_wgrad64_bf7(_p(x), _p(x2), _p(vec), _p(wpart), _p(dw7), _p(db), B, T, 2, 0, 1, _p(gsc7), st), F7, 2 * FR)
timeit("wgrad64bf7 k7 none x addvec", lambda: lib.wm_wgrad64_bf7(_p(x), _p(x2), _p(vec), _p(wpart), _p(dw7), _p(db), B, T, 2, 0, 0, None, st), F7, 2 * FR)
timeit("bn_add_relu", lambda: lib.wm_bn_add_relu(_p(x), _p(x2), _p(c[0]), _p(c[1]), _p(y), B, T, st), None, 3 * FR)
part = _f32(B * 128, device=dev)
timeit("relu_bwd_reduce", lambda: lib.wm_relu_bwd_reduce(_p(x), _p(x2), _p(x3), _p(y), _p(part), B, T, st), None, 4 * FR)
mask = torch.empty(B * 64 * ((T + 31) // 32), dtype=torch.int32, device=dev)
dzmax = _f32(B * 64, device=dev)
timeit("bn_add_relu_mask", lambda: lib.wm_bn_add_relu_mask(_p(x), _p(x2), _p(c[0]), _p(c[1]), _p(y), _p(mask), B, T, st), None, 3 * FR + mask.numel() * 4)
timeit("relu_bwd_reduce_mask sums only", lambda: lib.wm_relu_bwd_reduce_mask(_p(x), _p(mask), _p(x3), None, _p(part), _p(dzmax), B, T, st), None, 2 * FR + mask.numel() * 4)
# LSTM
wi = torch.randn(256, 64, device=dev) * 0.1; wh = torch.randn(256, 64, device=dev) * 0.1
bi = torch.randn(256, device=dev) * 0.1
xp = _f32(B, T, 256, device=dev); cst = _f32(B, T, 64, device=dev); h = torch.empty_like(x)
FL = 2.0 * 256 * 64 * T * B
timeit("lstm_xproj", lambda: lib.wm_lstm_xproj(_p(x), _p(wi), _p(bi), _p(bi), _p(xp), B, T, st), FL, FR + 4 * FR, n=3)
def lf():
    lib.wm_lstm_xproj(_p(x), _p(wi), _p(bi), _p(bi), _p(xp), B, T, st)
    lib.wm_lstm_fwd(_p(xp), _p(wh), _p(h), _p(xp), _p(cst), B, T, st)
timeit("lstm xproj+fwd(save)", lf, None, None, n=2)
gt = _f32(B, T, 256, device=dev)
timeit("lstm fused fwd (save)", lambda: lib.wm_lstm_fwd_fused(_p(x), _p(wi), _p(bi), _p(bi), _p(wh), _p(h), _p(gt), _p(cst), B, T, st), None, None, n=2)
timeit("lstm fused fwd (inference)", lambda: lib.wm_lstm_fwd_fused(_p(x), _p(wi), _p(bi), _p(bi), _p(wh), _p(h), None, None, B, T, st), None, None, n=2)
timeit("lstm_bwd (on stale gates)", lambda: lib.wm_lstm_bwd(_p(xp), _p(cst), _p(x2), _p(wh), B, T, st), None, None, n=2)
timeit("lstm bwd fused (+dx)", lambda: lib.wm_lstm_bwd_fused(_p(xp), _p(cst), _p(x2), _p(wh), _p(wi), _p(y), B, T, st), None, None, n=2)
timeit("lstm_dx", lambda: lib.wm_lstm_dx(_p(xp), _p(wi), _p(y), B, T, st), FL, 5 * FR, n=3)
lpart = _f32(256 * (256 * 128 + 256), device=dev); dwi = torch.empty_like(wi); dwh = torch.empty_like(wh); dbi = _f32(256, device=dev); dbh = _f32(256, device=dev)
timeit("lstm_wgrad", lambda: lib.wm_lstm_wgrad(_p(xp), _p(x), _p(h), _p(lpart), _p(dwi), _p(dwh), _p(dbi), _p(dbh), B, T, 0, st), 2 * FL, 6 * FR, n=3)
lpart2 = _f32(B * (256 * 128 + 256), device=dev)
timeit("lstm_bwd_wgrad (recurrence + helper waves: dW_ih dW_hh db)", lambda: lib.wm_lstm_bwd_wgrad(_p(xp), _p(cst), _p(x2), _p(wh), _p(x), _p(h), _p(lpart2), _p(dwi), _p(dwh), _p(dbi), _p(dbh), B, T, 0, st), None, None, n=2)
# heads / stem (Detector-side: 2B clips)
s = torch.randn(2 * B, 1, T, device=dev); ws = torch.randn(64, 1, 7, device=dev); X2 = torch.randn(2 * B, 64, T, device=dev); Y2 = torch.empty_like(X2)
timeit("stem_fwd (2B)", lambda: lib.wm_stem_fwd(_p(s), _p(ws), _p(bias), _p(Y2), 2 * B, T, st), None, 2 * FR)
sp = _f32(512 * 512, device=dev); dws = torch.empty_like(ws); ds = torch.empty_like(s)
timeit("stem_bwd +ds (2B)", lambda: lib.wm_stem_bwd(_p(X2), _p(s), _p(ws), _p(ds), _p(sp), _p(dws), _p(db), 2 * B, T, 2 * B, 0, st), None, 2 * FR)
timeit("stem_bwd, ds for the first half (2B: Detector)", lambda: lib.wm_stem_bwd(_p(X2), _p(s), _p(ws), _p(ds), _p(sp), _p(dws), _p(db), 2 * B, T, B, 0, st), None, 2 * FR)
timeit("stem_bwd, no ds (B: Generator)", lambda: lib.wm_stem_bwd(_p(X2), _p(s), _p(ws), None, _p(sp), _p(dws), _p(db), B, T, 0, 0, st), None, FR)
w17 = torch.randn(17, 64, 1, device=dev); b17 = torch.randn(17, device=dev); lg = _f32(2 * B, T, 17, device=dev)
timeit("headN_fwd (2B)", lambda: lib.wm_headN_fwd(_p(X2), _p(w17), _p(b17), _p(lg), 2 * B, T, 17, st), None, 2 * FR * 1.27)
hp = _f32(256 * (17 * 64 + 17), device=dev); dw17 = torch.empty_like(w17); db17 = _f32(17, device=dev)
timeit("headN_bwd (2B)", lambda: lib.wm_headN_bwd(_p(lg), _p(X2), _p(w17), _p(Y2), _p(hp), _p(dw17), _p(db17), 2 * B, T, 17, 0, st), None, 2 * FR * 2.27)
# ---- BCE over the (2B, T, 17) logits
lg = torch.randn(2 * B, T, 17, device=dev) * 3
msgs = torch.randint(0, 65536, (B,), device=dev)
bpart = _f32(2 * 2 * B * ((T * 17 + 4095) // 4096), device=dev); bout = torch.zeros(2, device=dev); dlg = torch.empty_like(lg)
gl = torch.ones(1, device=dev)
timeit("bce_fwd", lambda: lib.wm_bce_fwd(_p(lg), _p(msgs), _p(bpart), _p(bout[0]), _p(bout[1]), B, 2 * B, T, 17, st), None, 2 * B * T * 17 * 4.0)
timeit("bce_bwd", lambda: lib.wm_bce_bwd(_p(lg), _p(msgs), _p(gl), _p(gl), _p(dlg), B, 2 * B, T, 17, st), None, 2 * 2 * B * T * 17 * 4.0)
