import sys, time, torch
sys.path.insert(0, '.')
import bench
model = sys.argv[1] if len(sys.argv) > 1 else "main14b_2"
B = 128 if model == "main14b_2" else 256
dev = torch.device("cuda:0")
G, D, step, timers, bf = bench.build_workload(model, "train", B, 0, 1, dev, False, False)
for _ in range(3): step()
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    for _ in range(10): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{model}: host issue {1e3*(t1-t0)/10:.2f} ms/step, wall {1e3*(t2-t0)/10:.2f} ms/step", flush=True)
