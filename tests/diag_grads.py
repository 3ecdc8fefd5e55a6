"""diagnostic (not a test): per-parameter gradient error of HIP and of the fp32 CPU oracle against an fp64 CPU run"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import awm_amd
from oracle import recipes as R, wm_oracle as O
dev = torch.device("cuda:0")
B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 4, int(sys.argv[2]) if len(sys.argv) > 2 else 16000
gsd, dsd = R.reference_layout_init(); R.perturb_bn_(gsd, 7); R.perturb_bn_(dsd, 8)
s = O.synthetic_clips(B, seed=1235, T=T); msg = O.synthetic_messages(B, seed=4322)
def run(dtype):
    g2 = {k: (v.to(dtype) if v.is_floating_point() else v).clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in gsd.items()}
    d2 = {k: (v.to(dtype) if v.is_floating_point() else v).clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in dsd.items()}
    tot, out = O.step_losses(g2, d2, s.to(dtype), msg, training=True)
    tot.backward()
    return g2, d2, out
g32, d32, o32 = run(torch.float32)
g64, d64, o64 = run(torch.float64)
G, D = awm_amd.Generator(16), awm_amd.Detector(16)
G.load_state_dict(gsd); D.load_state_dict(dsd); G.to(dev).train(); D.to(dev).train()
total, out = awm_amd.forward_losses(G, D, s.to(dev), msg.to(dev)); total.backward()
def rel(a, ref): return float((a.double().cpu() - ref.double()).abs().max() / (ref.double().abs().max() + 1e-30))
for k in ("l1", "mel", "loud", "loc", "bce", "hf", "total"):
    print(f"{k:8s} hip-vs-64 {rel(out[k], o64[k]):.2e}   cpu32-vs-64 {rel(o32[k], o64[k]):.2e}")
for k in ("delta_raw", "delta", "logits"):
    print(f"{k:10s} hip-vs-64 {rel(out[k], o64[k]):.2e}   cpu32-vs-64 {rel(o32[k], o64[k]):.2e}")
for name, mod, r32, r64 in (("G", G, g32, g64), ("D", D, d32, d64)):
    for k, p in mod.named_parameters():
        print(f"{name}.{k:32s} hip-vs-64 {rel(p.grad, r64[k].grad):.2e}   cpu32-vs-64 {rel(r32[k].grad, r64[k].grad):.2e}   |g|max {float(r64[k].grad.abs().max()):.2e}")
