"""diagnostic (not a test): is the config-5 step host-bound?  host time to ENQUEUE a step vs wall time per step"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import awm_amd
from awm_amd import main14b_2 as M
import bench
dev = torch.device("cuda:0")
model = sys.argv[1] if len(sys.argv) > 1 else "main14b_2"
B = 128 if model == "main14b_2" else 256
torch.manual_seed(42)
if model == "main14b_2":
    G, D = M.Generator(hidden_dim=256).to(dev).train(), M.Detector().to(dev).train(); step_fn = M.train_step
else:
    G, D = awm_amd.Generator(16).to(dev).train(), awm_amd.Detector(16).to(dev).train(); step_fn = awm_amd.train_step
opt = awm_amd.FlatAdam([G, D], lr=1e-3)
s, msg = bench.synthetic_batch(B, 0, dev)
for _ in range(3): step_fn(G, D, opt, s, msg)
torch.cuda.synchronize()
enq = []
t0 = time.perf_counter()
for _ in range(10):
    a = time.perf_counter(); step_fn(G, D, opt, s, msg); enq.append(time.perf_counter() - a)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 10
print(f"{model}: wall {wall*1e3:.2f} ms/step, host enqueue {sum(enq)/10*1e3:.2f} ms/step (min {min(enq)*1e3:.2f}, max {max(enq)*1e3:.2f})")
awm_amd.ops.set_index_check("off")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): step_fn(G, D, opt, s, msg)
torch.cuda.synchronize()
print(f"   without the message-range check sync: wall {(time.perf_counter()-t0)/10*1e3:.2f} ms/step")
