"""Weight / buffer recipes shared by the golden-fixture generator and the tests.

TEST INFRASTRUCTURE ONLY (see oracle/wm_oracle.py header).  Fixtures store seeds and a
SHA-256 of the resulting state dict instead of 17 MB of embedding table.
"""
from __future__ import annotations

from collections import OrderedDict

import torch
import torch.nn as nn

WEIGHT_SEED = 42      # set_seed(42), py/main16.py:27
BN_SEED_G = 7
BN_SEED_D = 8


def _resblock_layers(ch):
    # creation order of ResBlock.__init__, py/main16.py:115-121
    return nn.Sequential(OrderedDict(block=nn.Sequential(
        nn.Conv1d(ch, ch, 3, padding=1), nn.BatchNorm1d(ch), nn.ReLU(),
        nn.Conv1d(ch, ch, 3, padding=1), nn.BatchNorm1d(ch))))


def reference_layout_init(message_bits: int = 16, seed: int = WEIGHT_SEED):
    """Default PyTorch init in the parameter-creation order of Generator.__init__ then
    Detector.__init__ (py/main16.py:129-147, 171-181) under manual_seed(seed).
    Returns (generator_state, detector_state) with the reference's key names."""
    torch.manual_seed(seed)
    g = nn.Module()
    g.encoder = nn.Sequential(nn.Conv1d(1, 64, 7, padding=3), _resblock_layers(64), _resblock_layers(64))
    g.lstm = nn.LSTM(64, 64, batch_first=True)
    if message_bits > 0:
        g.embedding = nn.Embedding(2 ** message_bits, 64)
    g.decoder = nn.Sequential(nn.ConvTranspose1d(64, 64, 7, padding=3), _resblock_layers(64), nn.Conv1d(64, 1, 1))
    d = nn.Module()
    d.model = nn.Sequential(nn.Conv1d(1, 64, 7, padding=3), _resblock_layers(64), _resblock_layers(64),
                            nn.Conv1d(64, 1 + message_bits, 1))
    gs = OrderedDict((k, v.detach().clone()) for k, v in g.state_dict().items())
    ds = OrderedDict((k, v.detach().clone()) for k, v in d.state_dict().items())
    return gs, ds


def perturb_bn_(sd, seed: int):
    """Make BatchNorm affine params and running statistics non-trivial (default init is
    weight 1 / bias 0 / mean 0 / var 1, which would leave eval-mode BN untested)."""
    g = torch.Generator().manual_seed(seed)
    for k in sd:
        if k.endswith("running_mean"):
            sd[k] = 0.05 * torch.randn(sd[k].shape, generator=g)
        elif k.endswith("running_var"):
            sd[k] = 0.6 + 0.8 * torch.rand(sd[k].shape, generator=g)
        elif k.endswith("num_batches_tracked"):
            sd[k] = torch.tensor(3, dtype=torch.int64)
        elif ".block.1." in k or ".block.4." in k:
            if k.endswith("weight"):
                sd[k] = 0.8 + 0.4 * torch.rand(sd[k].shape, generator=g)
            elif k.endswith("bias"):
                sd[k] = 0.05 * torch.randn(sd[k].shape, generator=g)
    return sd
