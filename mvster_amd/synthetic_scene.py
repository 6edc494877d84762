"""A small analytic multi-view scene for the fusion tests and benchmarks: depth maps of a tilted plane seen by the
DTU-like cameras of ``mvster_amd.synthetic`` (exact ray/plane intersection), so that geometric consistency holds to
rounding where the plane is visible and can be broken on purpose."""
import numpy as np

from .synthetic import make_cameras


def plane_depth_maps(nviews, H, W, seed=0, noise=0.0, outlier_frac=0.0):
    """-> (depths [N,H,W] f32, Ks [N,3,3] f32, Es [N,4,4] f32).  Plane n.X = c in world coordinates."""
    cams = make_cameras(nviews, H, W, batch=1, rotate=True, seed=seed)["stage4"][0]     # [N,2,4,4]
    rng = np.random.RandomState(seed)
    n = np.array([0.15, -0.1, 1.0])
    n /= np.linalg.norm(n)
    c = 650.0
    depths, Ks, Es = [], [], []
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)
    for v in range(nviews):
        E = cams[v, 0].astype(np.float64)
        K = cams[v, 1, :3, :3].astype(np.float64)
        R, t = E[:3, :3], E[:3, 3]
        rays = np.linalg.inv(K) @ np.stack([xs.ravel(), ys.ravel(), np.ones(H * W)])   # camera-space rays, z = 1
        # world point = R^T (d * ray - t);  n . X = c  ->  d = (c + n.R^T t) / (n.R^T ray)
        nr = n @ R.T
        d = (c + nr @ t) / (nr @ rays)
        d = d.reshape(H, W)
        if noise:
            d = d * (1 + noise * rng.randn(H, W))
        if outlier_frac:
            bad = rng.rand(H, W) < outlier_frac
            d = np.where(bad, d * (1 + 0.2 * rng.rand(H, W)), d)
        depths.append(d.astype(np.float32))
        Ks.append(K.astype(np.float32))
        Es.append(E.astype(np.float32))
    return np.stack(depths), np.stack(Ks), np.stack(Es)
