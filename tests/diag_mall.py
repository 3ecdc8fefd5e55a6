"""Diagnostic (not a test): does walking a ResBlock's backward in clip chunks keep the shared operands in the Infinity Cache?

The data gradient of conv2 and the weight gradient of conv2 read the same three [B,64,T] frames (dz2, y2, y1).  Launched
over all 512 clips one after the other the second launch re-reads 6.3 GB from HBM; launched per chunk of n clips
(data gradient of the chunk, then weight gradient of the chunk) the second read can hit the 256 MiB Infinity Cache.

run:  python tests/diag_mall.py
"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import awm_amd  # noqa: E402,F401
from awm_amd import ops  # noqa: E402
from awm_amd._lib import lib  # noqa: E402


def p(t):
    return None if t is None else t.data_ptr()


def main():
    dev = torch.device("cuda:0")
    B, T = 512, 16000
    torch.manual_seed(0)
    dz = torch.randn(B, 64, T, device=dev)
    y2 = torch.randn(B, 64, T, device=dev)
    y1 = torch.randn(B, 64, T, device=dev)
    out = torch.empty(B, 64, T, device=dev)
    k = torch.randn(4, 64, device=dev) * 0.1
    sc = torch.rand(64, device=dev) + 0.5
    sh = torch.randn(64, device=dev) * 0.1
    w = torch.randn(64, 64, 3, device=dev) * 0.05
    stats = torch.empty(256 * 128, device=dev)
    wpart = torch.empty(2 * 256 * (3 * 4096 + 64), device=dev)
    dw = torch.zeros(64, 64, 3, device=dev)
    db = torch.zeros(64, device=dev)
    wp = ops.pack_w64_bf(w, 1)
    st = torch.cuda.current_stream().cuda_stream

    def dgrad(b0, n):
        o = b0 * 64 * T * 4
        lib.wm_conv64_bf(p(dz) + o, p(y2) + o, p(wp), p(k[0]), p(k[1]), p(k[3]), None, p(y1) + o, p(sc), p(sh), p(out) + o, p(stats),
                         n, T, 3, 1, 0, st)

    def wgrad(b0, n, acc):
        o = b0 * 64 * T * 4
        lib.wm_wgrad64_bf(p(dz) + o, p(y2) + o, p(k[0]), p(k[1]), p(k[3]), p(y1) + o, p(sc), p(sh), p(wpart), p(dw), p(db), n, T, 3, 1,
                          acc, st)

    def timed(fn, reps=3):
        best = 1e9
        for _ in range(reps):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return best

    td = timed(lambda: dgrad(0, B))
    tw = timed(lambda: wgrad(0, B, 0))
    print(f"whole batch ({B} clips): dgrad {td:.3f} ms, wgrad {tw:.3f} ms, sum {td + tw:.3f} ms", flush=True)
    for n in (8, 16, 24, 32, 64, 128):
        def only_d():
            for b0 in range(0, B, n):
                dgrad(b0, min(n, B - b0))

        def only_w():
            for b0 in range(0, B, n):
                wgrad(b0, min(n, B - b0), 1 if b0 else 0)

        def pair():
            for b0 in range(0, B, n):
                m = min(n, B - b0)
                dgrad(b0, m)
                wgrad(b0, m, 1 if b0 else 0)

        def pair_rev():
            for b0 in range(0, B, n):
                m = min(n, B - b0)
                wgrad(b0, m, 1 if b0 else 0)
                dgrad(b0, m)
        a, b, c, d = timed(only_d), timed(only_w), timed(pair), timed(pair_rev)
        print(f"chunks of {n:3d} clips: dgrad only {a:.3f} ms, wgrad only {b:.3f} ms, (dgrad, wgrad) per chunk {c:.3f} ms, "
              f"(wgrad, dgrad) per chunk {d:.3f} ms", flush=True)
    # the element-wise pair of the forward: bn_add_relu writes `out`, the next block's conv1 reads it
    # hot re-read test: the same n clips again and again vs the whole batch
    for n in (8, 16, 32):
        t_hot = timed(lambda: [wgrad(0, n, 0) for _ in range(B // n)])
        print(f"wgrad of the SAME {n} clips x {B // n}: {t_hot:.3f} ms (vs {tw:.3f} ms over {B} different clips)", flush=True)
        t_hot = timed(lambda: [dgrad(0, n) for _ in range(B // n)])
        print(f"dgrad of the SAME {n} clips x {B // n}: {t_hot:.3f} ms (vs {td:.3f} ms)", flush=True)


if __name__ == "__main__":
    main()
