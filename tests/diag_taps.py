"""diagnostic: gradients at intermediate activations of G in a full train step, HIP vs fp64 CPU oracle"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import awm_amd
from oracle import recipes as R, wm_oracle as O
dev = torch.device("cuda:0")
B, T = 4, 16000
gsd, dsd = R.reference_layout_init(); R.perturb_bn_(gsd, 7); R.perturb_bn_(dsd, 8)
s = O.synthetic_clips(B, seed=1235, T=T); msg = O.synthetic_messages(B, seed=4322)
def run(dtype):
    g2 = {k: (v.to(dtype) if v.is_floating_point() else v).clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in gsd.items()}
    d2 = {k: (v.to(dtype) if v.is_floating_point() else v).clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in dsd.items()}
    taps = {}
    sd = s.to(dtype)
    delta_raw = O.generator_forward(g2, sd, msg, training=True, new_stats={}, taps=taps)
    taps["delta_raw"] = delta_raw
    for v in taps.values(): v.retain_grad()
    delta = O.postprocess(delta_raw); delta.retain_grad(); taps["delta"] = delta
    s_w = sd + delta
    import torch.nn.functional as F
    logits = O.detector_forward(d2, torch.cat([s_w, sd], 0), training=True, new_stats={})
    det, dec = logits[:, :, 0], logits[:B, :, 1:]
    tgt = torch.cat([torch.ones(B, T), torch.zeros(B, T)]).to(dtype)
    bits = O.message_bits_target(msg).to(dtype).unsqueeze(1).expand(-1, T, -1)
    loc = F.binary_cross_entropy_with_logits(det, tgt); bce = F.binary_cross_entropy_with_logits(dec, bits)
    total = delta.abs().mean() + 4 * O.mel_loss(sd, s_w) + 20 * O.loudness_loss(sd, s_w) + 10 * loc + bce + 5 * O.high_freq_penalty(delta)
    total.backward()
    return taps
t64 = run(torch.float64); t32 = run(torch.float32)
G, D = awm_amd.Generator(16), awm_amd.Detector(16)
G.load_state_dict(gsd); D.load_state_dict(dsd); G.to(dev).train(); D.to(dev).train()
caps = {}
def cap(name):
    def hook(mod, inp, out):
        out.retain_grad(); caps[name] = out
    return hook
G.encoder[0].register_forward_hook(cap("enc0")); G.encoder[2].register_forward_hook(cap("enc"))
G.decoder[1].register_forward_hook(cap("dec1")); G.decoder[2].register_forward_hook(cap("delta_raw"))
total, out = awm_amd.forward_losses(G, D, s.to(dev), msg.to(dev))
out["delta"].retain_grad(); caps["delta"] = out["delta"]
total.backward()
def rel(a, ref): return float((a.double().cpu() - ref.double()).abs().max() / (ref.double().abs().max() + 1e-30))
for k in ("delta", "delta_raw", "dec1", "enc", "enc0"):
    ga, g64, g32 = caps[k].grad, t64[k].grad, t32[k].grad
    err = (ga.double().cpu() - g64).abs(); sc = float(g64.abs().max())
    nbig = int((err > 1e-3 * sc).sum())
    print(f"grad@{k:10s} hip-vs-64 {rel(ga, g64):.2e}  cpu32-vs-64 {rel(g32, g64):.2e}  value hip-vs-64 {rel(caps[k], t64[k]):.2e}  n(err>1e-3)={nbig} of {err.numel()}")
    i = int(err.argmax()); print("     worst at", [int(v) for v in torch.unravel_index(torch.tensor(i), err.shape)], float(ga.flatten()[i]), float(g64.flatten()[i]))
