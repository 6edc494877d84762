"""Seeded synthetic multi-view inputs with the reference's input contract.

The contract is the one the reference's loaders produce and `MVS4net.forward`
consumes (reference datasets/dtu_yao4.py:139-194, models/MVS4Net.py:60):

* ``imgs``           list of N tensors ``[B,3,H,W]`` (view 0 = reference view)
* ``proj_matrices``  dict ``stage1..stage4`` -> ``[B,N,2,4,4]``; ``[:,:,0]`` is the
  4x4 extrinsic, ``[:,:,1,:3,:3]`` the intrinsic scaled to that stage
  (1/8, 1/4, 1/2, 1 of full resolution)
* ``depth_values``   ``[B,2]`` = (depth_min, depth_max)

Cameras are DTU-like: focal 2892.33/2883.18 px at 1600x1200
rescaled to HxW, 60 mm baselines, depth range 425 .. 425+192*2.5*1.06 mm.
Used by bench.py, tests and the golden-fixture generator; no file IO.
"""
import math

import numpy as np
import torch

DTU_DEPTH_MIN = 425.0
DTU_DEPTH_MAX = 425.0 + 192 * 2.5 * 1.06


def _rot_xyz(rx, ry, rz):
    cx, sx = math.cos(rx), math.sin(rx)
    cy, sy = math.cos(ry), math.sin(ry)
    cz, sz = math.cos(rz), math.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]], dtype=np.float64)
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], dtype=np.float64)
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]], dtype=np.float64)
    return Rz @ Ry @ Rx


def make_cameras(nviews, H, W, batch=1, rotate=True, seed=0):
    """Return proj_matrices dict of float32 numpy arrays ``[B,N,2,4,4]``."""
    rng = np.random.RandomState(seed)
    K = np.array([[2892.33 * W / 1600.0, 0.0, W / 2.0],
                  [0.0, 2883.18 * H / 1200.0, H / 2.0],
                  [0.0, 0.0, 1.0]], dtype=np.float64)
    base = np.zeros((batch, nviews, 2, 4, 4), dtype=np.float64)
    for b in range(batch):
        for v in range(nviews):
            E = np.eye(4)
            if v > 0:
                # alternate left/right/up/down of the reference camera, 60 mm steps
                step = 60.0 * ((v + 1) // 2)
                sign = -1.0 if v % 2 else 1.0
                tx = sign * step
                ty = 12.0 * (v - nviews / 2.0) + 3.0 * b
                tz = 2.0 * v
                if rotate:
                    ang = rng.uniform(-0.03, 0.03, size=3)
                    # toe-in so that the views overlap around mid depth
                    ang[1] += math.atan2(tx, 0.5 * (DTU_DEPTH_MIN + DTU_DEPTH_MAX)) * 0.9
                    E[:3, :3] = _rot_xyz(*ang)
                E[:3, 3] = [-tx, -ty, tz]
            base[b, v, 0] = E
            base[b, v, 1, :3, :3] = K
    out = {}
    for s, f in enumerate((0.125, 0.25, 0.5, 1.0)):
        m = base.copy()
        m[:, :, 1, :2, :] *= f
        out["stage%d" % (s + 1)] = m.astype(np.float32)
    return out


def make_inputs(nviews=5, H=512, W=640, batch=1, seed=0, device="cpu", rotate=True):
    """Seeded inputs for one forward: (imgs, proj_matrices, depth_values)."""
    g = torch.Generator().manual_seed(seed)
    imgs = [torch.rand(batch, 3, H, W, generator=g) for _ in range(nviews)]
    # low-frequency structure so that features are not pure white noise
    for v in range(nviews):
        yy = torch.linspace(0, 6.0 + v, H).view(1, 1, H, 1)
        xx = torch.linspace(0, 9.0 - v, W).view(1, 1, 1, W)
        imgs[v] = (0.5 * imgs[v] + 0.25 * (torch.sin(xx + v) * torch.cos(yy) + 1.0)).contiguous()
    cams = make_cameras(nviews, H, W, batch=batch, rotate=rotate, seed=seed)
    proj = {k: torch.from_numpy(v).to(device) for k, v in cams.items()}
    depth_values = torch.tensor([[DTU_DEPTH_MIN, DTU_DEPTH_MAX]] * batch, dtype=torch.float32)
    imgs = [i.to(device) for i in imgs]
    return imgs, proj, depth_values.to(device)


def randomize_state(state_dict, seed=0, prob_gain=20.0, feat_gain=6.0):
    """Return a copy of ``state_dict`` with non-trivial BatchNorm statistics and
    a sharpened ``prob`` head, keyed only by parameter *names* so that it applies
    to any module tree using the reference's state_dict layout.

    Default-initialised BN (mean 0, var 1, gamma 1, beta 0) would let a kernel
    that ignores the statistics pass; near-uniform softmax outputs of a
    random-init ``prob`` layer make argmax parity ill-posed.
    """
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k in sorted(state_dict.keys()):
        v = state_dict[k].detach().clone()
        if k.endswith("running_mean"):
            v = 0.1 * torch.randn(v.shape, generator=g)
        elif k.endswith("running_var"):
            v = 0.5 + torch.rand(v.shape, generator=g)
        elif k.endswith("num_batches_tracked"):
            pass
        elif (".bn." in k or k.split(".")[-2].isdigit()) and k.endswith("weight") and v.dim() == 1:
            v = 0.8 + 0.4 * torch.rand(v.shape, generator=g)
        elif (".bn." in k or k.split(".")[-2].isdigit()) and k.endswith("bias") and v.dim() == 1 and \
                k.replace("bias", "running_mean") in state_dict:
            v = 0.1 * torch.randn(v.shape, generator=g)
        elif k.endswith("prob.weight"):
            v = v * prob_gain
        elif k.startswith("feature.out") and k.endswith("weight"):
            v = v * feat_gain      # O(1) features -> O(1) correlations -> realistic softmax margins
        out[k] = v
    return out
