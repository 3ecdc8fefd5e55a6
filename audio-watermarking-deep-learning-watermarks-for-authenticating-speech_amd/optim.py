"""Flat-buffer parameter store + fused Adam for the step either side of the hot path
(torch.optim.Adam(list(G.parameters()) + list(D.parameters()), lr=1e-3), py/main16.py:504, :278).

All parameters of the given modules are re-pointed at views of ONE contiguous fp32 buffer, and so are
their .grad tensors.  That gives (a) a single-launch Adam update (wm_adam_step) instead of 85 small
tensors, and (b) a ready-made single bucket for the data-parallel gradient all-reduce (distributed.py).
state_dict()/load_state_dict() of the modules keep working: the views ARE the parameters.
"""
from __future__ import annotations

import torch

from . import ops
from ._lib import lib


class FlatAdam:
    def __init__(self, modules, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, overlap_wgrad=False):
        self.params = [p for m in modules for p in m.parameters()]
        if not self.params:
            raise ValueError("no parameters")
        dev = self.params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FlatAdam drives a HIP kernel: move the modules to the GPU first (no CPU fallback)")
        n = sum(p.numel() for p in self.params)
        self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            k = p.numel()
            self.flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + k].view_as(p)
            p.grad = self.grad[off:off + k].view_as(p)
            p._wm_grad = p.grad if overlap_wgrad else None     # destination for the side-stream weight-gradient GEMMs
            off += k
        ops.set_async_wgrad(overlap_wgrad)
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.t = 0

    def zero_grad(self, set_to_none=False):
        self.grad.zero_()          # the views stay attached; autograd accumulates into them in place

    def finish_backward(self):
        """join the weight-gradient side stream (call after backward, before reading / all-reducing self.grad)"""
        ops.join_side_stream()

    def step(self):
        self.finish_backward()
        self.t += 1
        lib.wm_adam_step(self.flat.data_ptr(), self.grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), self.flat.numel(),
                         self.lr, self.betas[0], self.betas[1], self.eps, self.t, torch.cuda.current_stream().cuda_stream)
