"""diagnostic: per-step wall time of the main16 B=256 train step from process start on a fresh box (is the first second slow?)"""
import sys, time, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t_start = time.perf_counter()
import bench
dev = torch.device("cuda:0")
G, D, step, timers, bf = bench.build_workload("main16", "train", 256, 0, 1, dev, False, False)
torch.cuda.synchronize()
print(f"build {time.perf_counter() - t_start:.1f} s", flush=True)
import gc
if os.environ.get("WM_DIAG_GC") == "0":
    gc.collect(); gc.freeze(); gc.disable()
def seg():
    st = torch.cuda.memory_stats()
    return st["segment.all.allocated"], st["num_alloc_retries"], st["reserved_bytes.all.current"] >> 20
ts = []
s0 = seg()
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 100):
    t0 = time.perf_counter(); step(); torch.cuda.synchronize(); dt = 1e3 * (time.perf_counter() - t0); ts.append(dt)
    if dt > 70 and i > 2:
        print(f"step {i}: {dt:.0f} ms, segments / retries / reserved MiB {seg()} (at start {s0}), gc counts {gc.get_count()} stats {[g['collections'] for g in gc.get_stats()]}", flush=True)
print(" ".join(f"{t:.0f}" for t in ts), flush=True)
# then unsynchronised blocks of 20 steps, as bench times them
for rep in range(4):
    t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize()
    print(f"block of 20: {1e3 * (time.perf_counter() - t0) / 20:.2f} ms/step", flush=True)
