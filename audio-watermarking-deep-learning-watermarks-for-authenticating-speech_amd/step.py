"""The step recipe of the reference's hot loop (train_one_epoch / validate_one_epoch / evaluate_model,
py/main16.py:238-278, :312-347, :377-403), expressed over the HIP modules.  Host glue only."""
from __future__ import annotations

from collections import OrderedDict

import torch

from . import losses as L

# py/main16.py:38-43
LOSS_WEIGHTS = OrderedDict(l1=1.0, mel=4.0, loud=20.0, loc=10.0, bce=1.0, hf=5.0)

_mel = L.MultiScaleMelLoss()
_loud = L.TFLoudnessLoss()


def forward_losses(generator, detector, s, message):
    """delta -> post-processing -> detector on cat([s_w, s]) -> the six loss terms and both totals."""
    B = s.shape[0]
    delta_raw = generator(s, message)                                  # :244
    delta = L.postprocess(delta_raw)                                   # :245-247
    s_w = s + delta                                                    # :248
    logits = detector(torch.cat([s_w, s], dim=0), input_grad_rows=s.shape[0])                    # :249-250
    loc, bce = L.detection_losses(logits, message)                     # :252-264
    l1 = L.l1_to_zero(delta)                                           # :266
    mel = _mel(s, s_w)                                                 # :267
    loud = _loud(s, s_w)                                               # :268
    hf = L.high_freq_penalty(delta)                                    # :271
    raw = l1 + mel + loud + loc + bce                                  # :273
    w = LOSS_WEIGHTS
    total = w["l1"] * l1 + w["mel"] * mel + w["loud"] * loud + w["loc"] * loc + w["bce"] * bce + w["hf"] * hf   # :275-276
    return total, OrderedDict(delta_raw=delta_raw, delta=delta, s_w=s_w, logits=logits, l1=l1, mel=mel, loud=loud, loc=loc,
                              bce=bce, hf=hf, raw_total=raw, total=total)


def train_step(generator, detector, optimizer, s, message, grad_sync=None):
    """One iteration of train_one_epoch's loop body (:242-278): zero_grad, forward, backward, optimizer step.
    `grad_sync` (optional callable) runs between backward and the update -- the data-parallel all-reduce."""
    optimizer.zero_grad(set_to_none=not hasattr(optimizer, "flat"))
    if hasattr(grad_sync, "begin_step"):
        grad_sync.begin_step()
    from . import ops
    try:
        with ops.index_check_mode("deferred" if ops._CHECK_INDEX["mode"] == "sync" else ops._CHECK_INDEX["mode"]):
            total, out = forward_losses(generator, detector, s, message)   # no mid-step sync for the message-id range check ...
        total.backward()
        if hasattr(optimizer, "finish_backward"):
            optimizer.finish_backward()
        if grad_sync is not None:
            grad_sync()
        # ... its flag (copied to pinned memory right after the lookup) has landed long before backward returns: read it here, so
        # that a bad id raises BEFORE the update, as nn.Embedding's IndexError does (py/main16.py:158), never one step late
        ops.check_message_ids(wait=True, what="this train_step's batch")
    except BaseException:
        ops.drop_pending_message_checks()        # a step that died half-way must not report its flag inside a later, valid step
        raise
    optimizer.step()
    return out


@torch.no_grad()
def eval_forward(generator, detector, s, message):
    """evaluate_model's per-batch quantities (:383-403).  In eval mode BatchNorm uses running statistics, so the Detector's
    rows are independent: the clean half D(s) does not wait for the Generator -- it is queued for the side stream and released
    when the Generator reaches its latency-bound LSTM (B clips keep only B of the 256 CUs busy there), and the two halves are
    concatenated afterwards (bit-identical to the single 2B-row call)."""
    from . import ops
    B = s.shape[0]
    overlap = (not generator.training) and (not detector.training) and s.is_cuda
    box = {}
    if overlap:
        def clean_half():
            box["lg"] = detector(s)
        ops._on_side((s,), clean_half)                # queued: released on the side stream when the LSTM launch is reached
    delta = L.postprocess(generator(s, message))
    if overlap:
        lg_wm = detector(s + delta)
        ops.join_side_stream()                        # (also releases the queue if the Generator had no LSTM call)
        box["lg"].record_stream(torch.cuda.current_stream())
        logits = torch.cat([lg_wm, box["lg"]], dim=0)
    else:
        logits = detector(torch.cat([s + delta, s], dim=0))
    probs = torch.sigmoid(logits[:, :, 0]).mean(dim=1)
    decoded = (torch.sigmoid(logits[:B, :, 1:]) > 0.5).float().mean(dim=1) > 0.5
    bits = ((message.unsqueeze(1) & (1 << torch.arange(logits.shape[-1] - 1, device=s.device))) > 0)
    return OrderedDict(delta=delta, logits=logits, prob_watermarked=probs[:B], prob_clean=probs[B:],
                       bit_accuracy=(decoded == bits).float().mean(dim=1), delta_rms=torch.sqrt((delta ** 2).mean(dim=[1, 2])))
