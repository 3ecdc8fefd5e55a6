"""On-disk formats a drop-in has to read and write: best-model state_dict files whose keys may
carry torch.compile's `_orig_mod.` prefix (py/main16.py:553-554, :707-712) and the resumable training dict of
py/main14d.py:540-572 {epoch, step, best_val, gen, det, opt, sched}."""
from __future__ import annotations

import os

import torch

PREFIX = "_orig_mod."


def strip_prefix(state_dict, prefix=PREFIX):
    return {(k[len(prefix):] if k.startswith(prefix) else k): v for k, v in state_dict.items()}


def save_best(generator, detector, gen_path="generator_best.pth", det_path="detector_best.pth", compile_prefix=False):
    """write the two state_dict files of py/main16.py:553-554 (optionally with the prefix the reference's compiled
    models produce, so files are interchangeable in both directions)"""
    for m, p in ((generator, gen_path), (detector, det_path)):
        sd = m.state_dict()
        if compile_prefix:
            sd = {PREFIX + k: v for k, v in sd.items()}
        torch.save(sd, p)


def load_best(model, path, map_location="cpu"):
    sd = torch.load(path, map_location=map_location, weights_only=True)
    return model.load_state_dict(strip_prefix(sd), strict=False)


def save_resumable(path, epoch, step, best_val, generator, detector, optimizer=None, scheduler=None):
    """py/main14d.py:540-558 layout"""
    from . import ops
    ops.check_message_ids()          # a deferred message-id check must not outlive the weights it would have corrupted
    ck = {"epoch": epoch, "step": step, "best_val": best_val, "gen": generator.state_dict(), "det": detector.state_dict(),
          "opt": _opt_state(optimizer), "sched": scheduler.state_dict() if scheduler is not None else None}
    tmp = path + ".tmp"
    torch.save(ck, tmp)
    os.replace(tmp, path)


def load_resumable(path, generator, detector, optimizer=None, scheduler=None, map_location="cpu"):
    """py/main14d.py:563-572; returns (epoch, step, best_val).  weights_only: nothing from the file is executed."""
    ck = torch.load(path, map_location=map_location, weights_only=True)
    generator.load_state_dict(strip_prefix(ck["gen"]))
    detector.load_state_dict(strip_prefix(ck["det"]))
    if optimizer is not None and ck.get("opt") is not None:
        _load_opt_state(optimizer, ck["opt"])
    if scheduler is not None and ck.get("sched") is not None:
        scheduler.load_state_dict(ck["sched"])
    return ck.get("epoch", 0), ck.get("step", 0), ck.get("best_val", float("inf"))


def _opt_state(opt):
    """`optimizer.state_dict()` (py/main14d.py:547) with its tensors copied to the CPU.  torch.optim's state_dict() hands out
    the LIVE per-parameter dicts (`sd["state"][i] is opt.state[p]`), so the copy is built from new dicts -- writing the CPU
    tensors into the returned ones would move a running torch.optim.Adam's moments off the GPU."""
    if opt is None:
        return None
    sd = opt.state_dict()
    state = {i: {k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in st.items()} for i, st in sd.get("state", {}).items()}
    return {"state": state, "param_groups": [dict(g) for g in sd.get("param_groups", [])]}


def _load_opt_state(opt, state):
    if state.get("flat_adam"):          # round-1 private schema {flat_adam, m, v, t, lr}: still readable
        if not hasattr(opt, "flat"):
            raise ValueError("checkpoint holds a round-1 FlatAdam state, the optimizer is not a FlatAdam")
        opt.m.copy_(state["m"]); opt.v.copy_(state["v"]); opt.t = int(state["t"]); opt.lr = float(state["lr"])
        return
    opt.load_state_dict(state)
