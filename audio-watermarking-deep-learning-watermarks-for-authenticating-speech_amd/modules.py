"""Drop-in nn.Module mirrors of the reference's hot-path classes (py/main16.py:112-186).

Same class names, constructor signatures, sub-module names and state_dict layout (SURVEY.md
appendix A), so checkpoints, optimizers and callers of the reference keep working; the
forward/backward arithmetic is the HIP path in ops.py.  The torch.nn layers created in
__init__ are used as *parameter containers only* (this also reproduces the reference's
default initialisation bit-for-bit under the same seed) -- their own forward is never run.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops


class ResBlock(nn.Module):
    """py/main16.py:112-125."""

    def __init__(self, ch):
        super().__init__()
        if ch != 64:
            raise ValueError("the MI355X hot path is built for the reference's 64-channel ResBlock")
        self.block = nn.Sequential(
            nn.Conv1d(ch, ch, 3, padding=1),
            nn.BatchNorm1d(ch),
            nn.ReLU(),
            nn.Conv1d(ch, ch, 3, padding=1),
            nn.BatchNorm1d(ch),
        )
        self.relu = nn.ReLU()

    def forward(self, x):
        c1, n1, _, c2, n2 = self.block
        return ops.ResBlockFn.apply(x, c1.weight, c1.bias, n1.weight, n1.bias, c2.weight, c2.bias, n2.weight, n2.bias,
                                    n1.running_mean, n1.running_var, n1.num_batches_tracked,
                                    n2.running_mean, n2.running_var, n2.num_batches_tracked, self.training)


def _rb_args(m):
    c1, n1, _, c2, n2 = m.block
    return (c1.weight, c1.bias, n1.weight, n1.bias, c2.weight, c2.bias, n2.weight, n2.bias,
            n1.running_mean, n1.running_var, n1.num_batches_tracked, n2.running_mean, n2.running_var, n2.num_batches_tracked)


def resblock_pair(m1, m2, x):
    """m2(m1(x)) for two ResBlocks in a row (py/main16.py:135-136, :178-179).  In a training step on the fused path the pair is
    ONE tape node (ops.ResBlockPairFn: the second block's backward also does the first block's ReLU backward and BatchNorm sums);
    otherwise -- inference, no gradients wanted, odd clip lengths, any of the knobs off, forward hooks on the blocks -- two calls."""
    c = ops._CONV
    if (m1.training and m2.training and torch.is_grad_enabled() and x.is_cuda and x.dim() == 3 and x.shape[-1] % 64 == 0
            and c["bf16x6"] and c["fused_bwd"] and c["mask_on_load"] and c["pair_fold"] and not ops._ASYNC["on"]
            and not m1._forward_hooks and not m2._forward_hooks and not m1._forward_pre_hooks and not m2._forward_pre_hooks):
        return ops.ResBlockPairFn.apply(x, *_rb_args(m1), *_rb_args(m2), True)
    return m2(m1(x))


class _Stem(nn.Conv1d):
    def forward(self, x, grad_rows=None):
        return ops.StemFn.apply(x, self.weight, self.bias, grad_rows)


class _Head1(nn.Conv1d):
    def forward(self, x):
        return ops.Head1Fn.apply(x, self.weight, self.bias)


class Generator(nn.Module):
    """Encoder -> LSTM -> (+ message embedding) -> decoder, py/main16.py:128-162."""

    def __init__(self, message_bits=0):
        super().__init__()
        self.message_bits = message_bits
        self.encoder = nn.Sequential(_Stem(1, 64, 7, padding=3), ResBlock(64), ResBlock(64))
        self.lstm = nn.LSTM(64, 64, batch_first=True)
        if message_bits > 0:
            self.embedding = nn.Embedding(2 ** message_bits, 64)
        self.decoder = nn.Sequential(nn.ConvTranspose1d(64, 64, 7, padding=3), ResBlock(64), _Head1(64, 1, 1))

    def forward(self, s, message=None):
        x = resblock_pair(self.encoder[1], self.encoder[2], self.encoder[0](s))  # (B,64,T)
        x = ops.LSTMFn.apply(x, self.lstm.weight_ih_l0, self.lstm.weight_hh_l0, self.lstm.bias_ih_l0, self.lstm.bias_hh_l0)
        vec = None
        if self.message_bits > 0 and message is not None:
            if message.dim() != 1 or message.shape[0] != s.shape[0]:
                raise ValueError(f"message must have shape ({s.shape[0]},), got {tuple(message.shape)}")
            vec = ops.EmbedFn.apply(self.embedding.weight, message.to(torch.int64))
        ct = self.decoder[0]
        x = ops.ConvT7Fn.apply(x, vec, ct.weight, ct.bias)
        x = self.decoder[1](x)
        return self.decoder[2](x)                                               # delta (B,1,T)


class _HeadN(nn.Conv1d):
    def forward(self, x):
        return ops.HeadNFn.apply(x, self.weight, self.bias)


class Detector(nn.Module):
    """Sample-level logits (B,T,1+bits), py/main16.py:170-186."""

    def __init__(self, message_bits=0):
        super().__init__()
        self.message_bits = message_bits
        output_dim = 1 + message_bits
        self.model = nn.Sequential(_Stem(1, 64, kernel_size=7, padding=3), ResBlock(64), ResBlock(64),
                                   _HeadN(64, output_dim, kernel_size=1))

    def forward(self, x, input_grad_rows=None):
        """`input_grad_rows` (optional, not in the reference): only clips [0, input_grad_rows) get an input gradient --
        the train step feeds [watermarked; clean] (py/main16.py:249) and the clean half is data."""
        x = self.model[0](x) if input_grad_rows is None else self.model[0](x, input_grad_rows)
        return self.model[3](resblock_pair(self.model[1], self.model[2], x))


def load_state_dict_strip_prefix(model, state_dict, prefix="_orig_mod."):
    """py/main16.py:707-712: accept checkpoints saved from torch.compile-wrapped models."""
    cleaned = {(k[len(prefix):] if k.startswith(prefix) else k): v for k, v in state_dict.items()}
    return model.load_state_dict(cleaned, strict=False)
