"""``FusedAdam``: torch.optim.Adam's update (L2 weight decay, bias correction, no amsgrad) for all parameters of a group
in two or three launches of one gfx950 kernel (``mvster_fused_adam``) instead of torch's multi-tensor kernels (6 launches,
0.21 ms for the 348 tensors of MVS4net inside the captured training step).  Drop-in for the reference's
``optim.Adam(params, lr=..., betas=(0.9, 0.999), weight_decay=...)`` (train_mvs4.py:367); ``state_dict()`` has Adam's
layout (per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq``), so checkpoints move between the two.

The step counter and the learning rate live on the device: a captured step (``graph.GraphedTrainStep``) keeps counting, and a
learning-rate schedule reaches it -- ``sync_hyperparameters()`` (called by ``GraphedTrainStep`` before every replay, and by
``step()``) rewrites the device cell when ``group["lr"]`` changed.  betas / eps / weight_decay are launch arguments: they are
frozen into a captured step, as with torch's own optimizers.
"""
import ctypes

import torch

from . import _lib


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, capturable=True,
                 fused=None, foreach=None, maximize=False, differentiable=False):
        if amsgrad or maximize or differentiable:
            raise NotImplementedError("FusedAdam: amsgrad / maximize / differentiable are not built (the reference's "
                                      "optimizer uses none of them, train_mvs4.py:367)")
        if not 0.0 <= lr or not 0.0 <= eps or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or weight_decay < 0.0:
            raise ValueError("FusedAdam: invalid hyper-parameters")
        # (capturable / fused / foreach: accepted for signature compatibility; this optimizer is always capturable)
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=False,
                                      capturable=True))
        self._flat = {}          # id(group) -> buffers and tables

    # ------------------------------------------------------------------ state
    def _init_group(self, group):
        ps = [p for p in group["params"] if p.requires_grad]
        if not ps:
            return None
        dev = ps[0].device
        for p in ps:
            if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous() or p.device != dev:
                raise RuntimeError("FusedAdam: contiguous float32 parameters on one GPU expected (the HIP path has no CPU fallback)")
            if p.numel() >= 1 << 31:
                raise RuntimeError("FusedAdam: tensors of 2^31 elements or more are not supported")
        sizes = [p.numel() for p in ps]
        offs, total = [], 0
        for n in sizes:
            offs.append(total)
            total += (n + 3) // 4 * 4
        f = {"params": ps, "n": len(ps),
             "exp_avg": torch.zeros(total, device=dev), "exp_avg_sq": torch.zeros(total, device=dev),
             "step_cells": torch.zeros(2, device=dev), "lr_cell": torch.full((1,), float(group["lr"]), device=dev),
             "lr_host": float(group["lr"]),
             "sizes": (ctypes.c_int * len(ps))(*sizes), "offs": (ctypes.c_int * len(ps))(*offs)}
        step_view = f["step_cells"][0]
        for p, n, o in zip(ps, sizes, offs):
            st = self.state[p]
            # an Adam state_dict loaded before the first step: adopt its moments
            if "exp_avg" in st and st["exp_avg"].numel() == n:
                f["exp_avg"][o:o + n].copy_(st["exp_avg"].reshape(-1))
                f["exp_avg_sq"][o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
                if "step" in st:
                    f["step_cells"][0] = float(st["step"])
            st["step"] = step_view
            st["exp_avg"] = f["exp_avg"][o:o + n].view(p.shape)
            st["exp_avg_sq"] = f["exp_avg_sq"][o:o + n].view(p.shape)
        self._flat[id(group)] = f
        return f

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._flat.clear()               # the loaded per-parameter tensors are adopted by the next _init_group

    def sync_hyperparameters(self):
        """Bring the device-side learning-rate cells up to ``group["lr"]`` (a scheduler's work between two steps)."""
        for group in self.param_groups:
            f = self._flat.get(id(group))
            if f is not None and float(group["lr"]) != f["lr_host"]:
                f["lr_host"] = float(group["lr"])
                f["lr_cell"].fill_(f["lr_host"])

    # ------------------------------------------------------------------ update
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        from .ops import _stream
        lib = _lib.load()
        for group in self.param_groups:
            f = self._flat.get(id(group)) or self._init_group(group)
            if f is None:
                continue
            if not torch.cuda.is_current_stream_capturing() and float(group["lr"]) != f["lr_host"]:
                f["lr_host"] = float(group["lr"])
                f["lr_cell"].fill_(f["lr_host"])
            ps = f["params"]
            grads = [p.grad for p in ps]
            if any(g is None for g in grads):
                if all(g is None for g in grads):
                    continue
                raise NotImplementedError("FusedAdam: some parameters of a group have no gradient (one step counter per "
                                          "group): put parameters that are not trained every step into a group of their own")
            for p, g in zip(ps, grads):
                if g.dtype != torch.float32 or not g.is_contiguous() or g.is_sparse or g.shape != p.shape:
                    raise RuntimeError("FusedAdam: dense contiguous float32 gradients expected")
            n = f["n"]
            pp = (ctypes.c_void_p * n)(*[p.data_ptr() for p in ps])
            gp = (ctypes.c_void_p * n)(*[g.data_ptr() for g in grads])
            b1, b2 = group["betas"]
            rc = lib.mvster_fused_adam(ctypes.cast(pp, ctypes.c_void_p), ctypes.cast(gp, ctypes.c_void_p),
                                       ctypes.cast(f["sizes"], ctypes.c_void_p), ctypes.cast(f["offs"], ctypes.c_void_p), n,
                                       f["exp_avg"].data_ptr(), f["exp_avg_sq"].data_ptr(), f["step_cells"].data_ptr(),
                                       f["lr_cell"].data_ptr(), float(b1), float(b2), float(group["eps"]),
                                       float(group["weight_decay"]), _stream())
            _lib.check(rc, "fused_adam")
        return loss
