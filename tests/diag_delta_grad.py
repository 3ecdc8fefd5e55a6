"""diagnostic: gradient w.r.t. delta_raw of each loss term (HIP vs fp64 CPU oracle), with error location"""
import sys, os, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import awm_amd
from oracle import recipes as R, wm_oracle as O
dev = torch.device("cuda:0")
B, T = 4, 16000
gsd, dsd = R.reference_layout_init(); R.perturb_bn_(gsd, 7); R.perturb_bn_(dsd, 8)
s = O.synthetic_clips(B, seed=1235, T=T); msg = O.synthetic_messages(B, seed=4322)
with torch.no_grad():
    d_raw = O.generator_forward(gsd, s, msg, training=True, new_stats={})
D = awm_amd.Detector(16); D.load_state_dict(dsd); D.to(dev).train()
mel_h, loud_h = awm_amd.MultiScaleMelLoss(), awm_amd.TFLoudnessLoss()

def terms_hip(dr):
    delta = awm_amd.postprocess(dr); sw = s.to(dev) + delta
    logits = D(torch.cat([sw, s.to(dev)], 0))
    loc, bce = awm_amd.detection_losses(logits, msg.to(dev))
    return dict(l1=awm_amd.l1_to_zero(delta), mel=mel_h(s.to(dev), sw), loud=loud_h(s.to(dev), sw), loc=loc, bce=bce,
                hf=awm_amd.high_freq_penalty(delta))
def terms_cpu(dr, dtype):
    d2 = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in dsd.items()}
    sd = s.to(dtype)
    delta = O.postprocess(dr); sw = sd + delta
    logits = O.detector_forward(d2, torch.cat([sw, sd], 0), training=True, new_stats={})
    det, dec = logits[:, :, 0], logits[:B, :, 1:]
    tgt = torch.cat([torch.ones(B, T), torch.zeros(B, T)]).to(dtype)
    bits = O.message_bits_target(msg).to(dtype).unsqueeze(1).expand(-1, T, -1)
    return dict(l1=delta.abs().mean(), mel=O.mel_loss(sd, sw), loud=O.loudness_loss(sd, sw),
                loc=F.binary_cross_entropy_with_logits(det, tgt), bce=F.binary_cross_entropy_with_logits(dec, bits),
                hf=O.high_freq_penalty(delta))
for name in ("l1", "mel", "loud", "loc", "bce", "hf"):
    a = d_raw.clone().to(dev).requires_grad_(); terms_hip(a)[name].backward()
    b = d_raw.clone().double().requires_grad_(); terms_cpu(b, torch.float64)[name].backward()
    c = d_raw.clone().requires_grad_(); terms_cpu(c, torch.float32)[name].backward()
    ga, gb, gc = a.grad.cpu().double(), b.grad, c.grad.double()
    err = (ga - gb).abs(); i = int(err.argmax()); sc = float(gb.abs().max())
    print(f"{name:5s} hip-vs-64 {float(err.max())/sc:.2e} at flat idx {i} (clip {i//T}, t {i%T}); cpu32-vs-64 {float((gc-gb).abs().max())/sc:.2e}; |g|max {sc:.2e}; hip val {float(ga.flatten()[i]):.4e} ref {float(gb.flatten()[i]):.4e}")
    big = torch.nonzero(err.flatten() > 1e-3 * sc).flatten()
    print("      #elements with err > 1e-3*max:", big.numel(), "first few t:", [(int(j)//T, int(j)%T) for j in big[:12]])
