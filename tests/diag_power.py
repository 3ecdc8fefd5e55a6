"""diagnostic: what the chip reports (shader clock, socket power) while one kernel family runs back to back -- evidence for DESIGN.md
section 8's "limited by the clock the chip sustains".  python tests/diag_power.py {dwgrad|lstm|bnrelu}"""
import os, subprocess, sys, threading, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import awm_amd
from awm_amd import ops
from awm_amd.ops import _p, _stream, _f32, lib
what = sys.argv[1] if len(sys.argv) > 1 else "dwgrad"
dev = torch.device("cuda:0"); B, T = 256, 16000
x = torch.randn(B, 64, T, device=dev) * 0.5; x2 = torch.randn(B, 64, T, device=dev) * 0.5; x3 = torch.randn(B, 64, T, device=dev) * 0.5
y = torch.empty_like(x)
st = _stream()
if what == "dwgrad":
    w = torch.randn(64, 64, 3, device=dev) * 0.05
    wph = ops.pack_w64_h(w, 1)
    k = torch.rand(4, 64, device=dev); sc = torch.rand(64, device=dev) + 0.5; sh = torch.randn(64, device=dev) * 0.1
    stats = _f32(256 * 128, device=dev); wpart = _f32(256 * (3 * 4096 + 64), device=dev); dw = torch.empty_like(w); db = _f32(64, device=dev)
    gsc = torch.tensor([1.0, 1.0], device=dev)
    fn = lambda: lib.wm_dwgrad64_bf(_p(x), _p(x2), _p(k[0]), _p(k[1]), _p(k[3]), _p(wph), _p(x3), _p(sc), _p(sh), _p(x3), _p(sc), _p(sh), _p(y),
                                    _p(stats), _p(wpart), _p(dw), _p(db), B, T, 1, 1, 0, None, 1, _p(gsc), None, st)
elif what == "lstm":
    wi = torch.randn(256, 64, device=dev) * 0.1; wh = torch.randn(256, 64, device=dev) * 0.1; bi = torch.randn(256, device=dev) * 0.1
    h = torch.empty_like(x)
    fn = lambda: lib.wm_lstm_fwd_fused(_p(x), _p(wi), _p(bi), _p(bi), _p(wh), _p(h), None, None, B, T, st)
else:
    c = [torch.rand(64, device=dev) + 0.5 for _ in range(2)]
    fn = lambda: lib.wm_bn_add_relu(_p(x), _p(x2), _p(c[0]), _p(c[1]), _p(y), B, T, st)
fn(); torch.cuda.synchronize()
stop = False
def smi():
    while not stop:
        try:
            out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showpower", "-d", "0"], capture_output=True, text=True, timeout=5).stdout
            keep = [l.strip() for l in out.splitlines() if ("sclk" in l or "Power" in l or "mclk" in l)]
            print(" | ".join(keep), flush=True)
        except Exception as e:
            print("rocm-smi:", e, flush=True)
        time.sleep(0.5)
t = threading.Thread(target=smi); t.start()
t0 = time.time(); n = 0
while time.time() - t0 < 4.0:
    for _ in range(20): fn()
    torch.cuda.synchronize(); n += 20
stop = True; t.join()
print(f"{what}: {n} launches, {1e3 * (time.time() - t0) / n:.3f} ms per launch", flush=True)
