"""Flat-buffer parameter store + fused Adam for the step either side of the hot path
(torch.optim.Adam(list(G.parameters()) + list(D.parameters()), lr=1e-3), py/main16.py:504, :278;
driven by OneCycleLR in py/main14d.py:491-507 and checkpointed as `optimizer.state_dict()`, :540-558).

All parameters of the given modules are re-pointed at views of ONE contiguous fp32 buffer, and so are
their .grad tensors.  That gives (a) a single-launch Adam update (wm_adam_step) instead of 85 small
tensors, and (b) a ready-made single bucket for the data-parallel gradient all-reduce (distributed.py).
state_dict()/load_state_dict() of the modules keep working: the views ARE the parameters.

FlatAdam is a torch.optim.Optimizer: `param_groups[0]["lr"]` / `["betas"]` are read at every step(), so
torch.optim.lr_scheduler.* (OneCycleLR cycles lr AND beta1) drive it unchanged, and state_dict() /
load_state_dict() use torch.optim.Adam's own layout ({state: {i: {step, exp_avg, exp_avg_sq}},
param_groups: [...]}), so resumable checkpoints are interchangeable with the reference's in both directions.
"""
from __future__ import annotations

import torch

from . import ops
from ._lib import lib


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, modules, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, overlap_wgrad=False):
        params = [p for m in modules for p in m.parameters()]
        if not params:
            raise ValueError("no parameters")
        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FlatAdam drives a HIP kernel: move the modules to the GPU first (no CPU fallback)")
        if any(p.dtype != torch.float32 or p.device != dev for p in params):
            raise ValueError("FlatAdam: every parameter must be fp32 on the same GPU")
        # same defaults dict as torch.optim.Adam, so its state_dict()s load here and ours load there
        defaults = dict(lr=lr, betas=(float(betas[0]), float(betas[1])), eps=float(eps), weight_decay=0, amsgrad=False,
                        maximize=False, foreach=None, capturable=False, differentiable=False, fused=None,
                        decoupled_weight_decay=False)
        super().__init__(params, defaults)
        self.params = params
        n = sum(p.numel() for p in params)
        self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        self._spans = []
        off = 0
        for p in params:
            k = p.numel()
            self.flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + k].view_as(p)
            p.grad = self.grad[off:off + k].view_as(p)
            p._wm_grad = p.grad if overlap_wgrad else None     # destination for the side-stream weight-gradient GEMMs
            self._spans.append((off, k))
            off += k
        ops.set_async_wgrad(overlap_wgrad)
        self.t = 0

    # scalar conveniences (the scheduler-facing truth is param_groups[0])
    @property
    def lr(self):
        return float(self.param_groups[0]["lr"])

    @lr.setter
    def lr(self, value):
        self.param_groups[0]["lr"] = float(value)

    @property
    def betas(self):
        return tuple(float(b) for b in self.param_groups[0]["betas"])

    @property
    def eps(self):
        return float(self.param_groups[0]["eps"])

    def zero_grad(self, set_to_none=False):
        self.grad.zero_()          # the views stay attached; autograd accumulates into them in place

    def finish_backward(self):
        """join the weight-gradient side stream (call after backward, before reading / all-reducing self.grad)"""
        ops.join_side_stream()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self.finish_backward()
        g = self.param_groups[0]
        if g.get("weight_decay", 0) or g.get("amsgrad", False) or g.get("maximize", False):
            raise ValueError("FlatAdam implements torch.optim.Adam's default update only (no weight_decay / amsgrad / maximize)")
        self.t += 1
        b1, b2 = g["betas"]
        lib.wm_adam_step(self.flat.data_ptr(), self.grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), self.flat.numel(),
                         float(g["lr"]), float(b1), float(b2), float(g["eps"]), self.t, torch.cuda.current_stream().cuda_stream)
        return loss

    # ---- torch.optim.Adam's checkpoint layout (py/main14d.py:547, :556) ----------------------------------------
    def state_dict(self):
        state = {}
        if self.t > 0:
            for i, (off, k) in enumerate(self._spans):
                shape = self.params[i].shape
                state[i] = {"step": torch.tensor(float(self.t)),
                            "exp_avg": self.m[off:off + k].view(shape).clone(),
                            "exp_avg_sq": self.v[off:off + k].view(shape).clone()}
        group = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        group["params"] = list(range(len(self.params)))
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        groups = sd["param_groups"]
        ids = [i for g in groups for i in g["params"]]
        if len(ids) != len(self.params):
            raise ValueError(f"optimizer state holds {len(ids)} parameters, this FlatAdam has {len(self.params)}")
        if len(groups) != 1:
            raise ValueError("FlatAdam keeps one param group (as torch.optim.Adam over one parameter list does)")
        steps = set()
        self.m.zero_(); self.v.zero_()
        for pos, pid in enumerate(ids):
            st = sd["state"].get(pid)
            if st is None:
                continue
            off, k = self._spans[pos]
            if st["exp_avg"].numel() != k:
                raise ValueError(f"optimizer state of parameter {pid}: {tuple(st['exp_avg'].shape)} vs {tuple(self.params[pos].shape)}")
            self.m[off:off + k].copy_(st["exp_avg"].reshape(-1))
            self.v[off:off + k].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError(f"per-parameter step counts differ ({sorted(steps)}): not a state FlatAdam can hold")
        self.t = steps.pop() if steps else 0
        keep = self.param_groups[0]["params"]
        self.param_groups[0].update({k: v for k, v in groups[0].items() if k != "params"})
        self.param_groups[0]["params"] = keep
