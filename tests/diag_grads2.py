"""diagnostic (not a test): test_all_grads_vs_oracle's configuration, per-parameter distances HIP-fp64 / CPU32-fp64 / HIP-CPU32"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import awm_amd
from awm_amd import ops
from oracle import recipes as R, wm_oracle as O
dev = torch.device("cuda:0")
B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 3, int(sys.argv[2]) if len(sys.argv) > 2 else 4000
gsd, dsd = R.reference_layout_init(); R.perturb_bn_(gsd, R.BN_SEED_G); R.perturb_bn_(dsd, R.BN_SEED_D)
msg = O.synthetic_messages(B, seed=71)
for seed in range(70, 170):
    s = O.synthetic_clips(B, seed=seed, T=T)
    with torch.no_grad():
        f = O.fir_lowpass(O.generator_forward(gsd, s, msg, training=True))
    if float((f.abs() - 0.02).abs().min()) >= 1e-5 * float(f.abs().max()):
        break
def run(dtype):
    g2 = {k: (v.to(dtype) if v.is_floating_point() else v).clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in gsd.items()}
    d2 = {k: (v.to(dtype) if v.is_floating_point() else v).clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in dsd.items()}
    tot, out = O.step_losses(g2, d2, s.to(dtype), msg, training=True, g_stats={}, d_stats={})
    tot.backward()
    return g2, d2
g32, d32 = run(torch.float32); g64, d64 = run(torch.float64)
def rel(a, ref): return float((a.double().cpu() - ref.double()).abs().max() / (ref.double().abs().max() + 1e-30))
for mode in (True, False):
    ops.set_conv_bf16x6(mode)
    G, D = awm_amd.Generator(16), awm_amd.Detector(16)
    G.load_state_dict(gsd); D.load_state_dict(dsd); G.to(dev).train(); D.to(dev).train()
    total, out = awm_amd.forward_losses(G, D, s.to(dev), msg.to(dev)); total.backward()
    print("mode bf16x6 =", mode)
    for name, mod, r32, r64 in (("G", G, g32, g64), ("D", D, d32, d64)):
        for k, p in mod.named_parameters():
            a, b, c = rel(p.grad, r64[k].grad), rel(r32[k].grad, r64[k].grad), rel(p.grad, r32[k].grad)
            flag = " <<<" if a > max(2 * b, 3e-4) else ""
            print(f"  {name}.{k:30s} hip-64 {a:.2e}  cpu32-64 {b:.2e}  hip-cpu32 {c:.2e}  |g|max {float(r64[k].grad.abs().max()):.2e}{flag}")
