"""diagnostic (not a test): per-layer timing of the config-5 (main14b_2, hidden 256) convolution family at B=128.
Every Conv1d / ConvTranspose1d shape of the Generator and the Detector: forward, data gradient, weight gradient."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import awm_amd
from awm_amd import main14b_2 as M
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
only = sys.argv[2] if len(sys.argv) > 2 else ""

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

rows = []
def conv(name, NB, cin, cout, k, s, p, L):
    if only and only not in name: return
    x = torch.randn(NB, cin, L, device=dev); w = torch.randn(cout, cin, k, device=dev) * 0.1; b = torch.randn(cout, device=dev)
    Lout = (L + 2 * p - k) // s + 1
    g = torch.randn(NB, cout, Lout, device=dev)
    fl = 2.0 * NB * cin * cout * k * Lout
    by = 4.0 * NB * (cin * L + cout * Lout)
    tf = timeit(lambda: M._conv_fwd(x, w, b, s, p, 1))
    td = timeit(lambda: M._conv_dgrad(g, w, s, p, L))
    tw = timeit(lambda: M._conv_wgrad(g, x, w.shape, s, p, True))
    rows.append((name, NB, fl, by, tf, td, tw))

def convT(name, NB, cin, cout, st, L):
    if only and only not in name: return
    x = torch.randn(NB, cin, L, device=dev); w = torch.randn(cin, cout, 2 * st, device=dev) * 0.1; b = torch.randn(cout, device=dev)
    Lout = (L - 1) * st - 2 * (st // 2) + 2 * st
    g = torch.randn(NB, cout, Lout, device=dev)
    fl = 2.0 * NB * cin * cout * 2 * Lout
    by = 4.0 * NB * (cin * L + cout * Lout)
    xx = x.clone().requires_grad_(); ww = w.clone().requires_grad_(); bb = b.clone().requires_grad_()
    tf = timeit(lambda: M._gconvT(x, w, b, st))
    pad = st // 2
    wp = w.permute(1, 2, 0).reshape(cout * 2 * st, cin).contiguous()
    td = timeit(lambda: M._gconv_raw(g, wp, None, 2 * st, st, pad, cin, L, 1, 0, cin, L))
    def wg():
        gp = M._gather_taps(g, st, st, pad, L + 1, 1)
        return M._gwgrad_raw(x, gp, cout * st, L + 1, 0, w.shape, 2, 0, False, 2, st, 0)
    tw = timeit(wg)
    rows.append((name, NB, fl, by, tf, td, tw))

for tag, NB in (("G", B), ("D", 2 * B)):
    conv(f"{tag}.init 1>32 k7", NB, 1, 32, 7, 1, 3, 16000)
    ch, L = 32, 16000
    for i, st in enumerate((2, 4, 5, 8)):
        Lo = (L + 2 - 3) // st + 1
        conv(f"{tag}.enc{i}.conv1 {ch}>{2*ch} s{st}", NB, ch, 2 * ch, 3, st, 1, L)
        conv(f"{tag}.enc{i}.conv2 {2*ch}>{2*ch}", NB, 2 * ch, 2 * ch, 3, 1, 1, Lo)
        conv(f"{tag}.enc{i}.skip {ch}>{2*ch} s{st}", NB, ch, 2 * ch, 1, st, 0, L)
        ch, L = 2 * ch, Lo
conv("G.proj 512>256 k1", B, 512, 256, 1, 1, 0, 50)
conv("G.fce 256>128 k7", B, 256, 128, 7, 1, 3, 50)
for tag, NB, c0 in (("G", B, 128), ("D", 2 * B, 512)):
    ch, L = c0, 50
    for st in (8, 5, 4, 2):
        convT(f"{tag}.up{st} {ch}>{ch//2}", NB, ch, ch // 2, st, L)
        L = (L - 1) * st - 2 * (st // 2) + 2 * st; ch //= 2
        conv(f"{tag}.rb{ch}@{L}", NB, ch, ch, 3, 1, 1, L)
conv("G.final 8>1 k7", B, 8, 1, 7, 1, 3, 16008)
conv("D.final 32>17 k7", 2 * B, 32, 17, 7, 1, 3, 16008)
print(f"{'layer':28s} {'NB':>4s} {'GFLOP':>8s} {'MB':>8s} | {'fwd ms':>8s} {'TF/s':>6s} {'TB/s':>5s} | {'dgrad ms':>8s} {'TF/s':>6s} | {'wgrad ms':>8s} {'TF/s':>6s}")
tot = [0, 0, 0]
for name, NB, fl, by, tf, td, tw in rows:
    mult = 2 if (".rb" in name) else 1        # a residual block holds two such convolutions
    tot[0] += tf * mult; tot[1] += td * mult; tot[2] += tw * mult
    print(f"{name:28s} {NB:4d} {fl/1e9:8.2f} {by/1e6:8.1f} | {tf:8.3f} {fl/tf/1e9:6.1f} {by/tf/1e9:5.2f} | {td:8.3f} {fl/td/1e9:6.1f} | {tw:8.3f} {fl/tw/1e9:6.1f}")
print("sum over layers (rb x2): fwd %.1f ms, dgrad %.1f ms, wgrad %.1f ms" % tuple(tot))
