"""Depth-map filtering for fusion: the step after the forward pass.

Mirrors the reference's ``test_mvs4.py`` functions -- ``check_geometric_consistency`` (:313-328, built on
``reproject_with_depth`` :273-310) and the per-reference-view part of ``filter_depth`` (:352-407) -- on one fused
gfx950 kernel (``mvster_geo_filter``) instead of NumPy + ``cv2.remap`` per view pair.  Inputs may be NumPy arrays
(as in the reference) or torch tensors; NumPy in gives NumPy out.  The small camera-matrix algebra stays on the host
in NumPy float32, exactly as the reference computes it; everything per pixel runs on the GPU.
"""
import numpy as np
import torch

from . import _lib


def _np32(a):
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    return np.asarray(a, dtype=np.float32)


def view_matrices(ref_K, ref_E, src_Ks, src_Es):
    """-> (ref_mats [18], view_mats [NS,42]) float64 holding the reference's float32 products/inverses."""
    ref_K, ref_E = _np32(ref_K), _np32(ref_E)
    ref_mats = np.concatenate([np.linalg.inv(ref_K).reshape(-1), ref_K.reshape(-1)]).astype(np.float64)
    rows = []
    for K, E in zip(src_Ks, src_Es):
        K, E = _np32(K), _np32(E)
        a = np.matmul(E, np.linalg.inv(ref_E))[:3]          # reference -> source camera
        b = np.matmul(ref_E, np.linalg.inv(E))[:3]          # source -> reference camera
        rows.append(np.concatenate([a.reshape(-1), K.reshape(-1), np.linalg.inv(K).reshape(-1), b.reshape(-1)]))
    return ref_mats, np.stack(rows).astype(np.float64)


def _dev_depth(d, dev):
    if isinstance(d, torch.Tensor):
        return d.to(dev, torch.float32).contiguous()
    return torch.from_numpy(np.ascontiguousarray(d, dtype=np.float32)).to(dev)


def geometric_filter(ref_depth, ref_K, ref_E, src_depths, src_Ks, src_Es, per_view=False, pix_thres=1.0, rel_thres=0.01,
                     device="cuda:0"):
    """One launch for a reference view against all its source views.  -> dict of CUDA tensors:
    ``mask_sum`` [H,W] int32 (consistent views per pixel), ``depth_sum`` [H,W] (sum of their reprojected depths) and,
    with ``per_view``, ``view_mask`` [NS,H,W] bool, ``view_depth``, ``x_src``, ``y_src`` [NS,H,W]."""
    dev = torch.device(device)
    dref = _dev_depth(ref_depth, dev)
    if isinstance(src_depths, torch.Tensor):
        dsrc = src_depths.to(dev, torch.float32).contiguous()
    else:
        dsrc = torch.stack([_dev_depth(d, dev) for d in src_depths]).contiguous()
    H, W = dref.shape
    NS = dsrc.shape[0]
    if tuple(dsrc.shape) != (NS, H, W) or len(src_Ks) != NS or len(src_Es) != NS:
        raise RuntimeError("geometric_filter: inconsistent shapes")
    ref_mats, view_mats = view_matrices(ref_K, ref_E, src_Ks, src_Es)
    ref_mats = torch.from_numpy(ref_mats).to(dev)
    view_mats = torch.from_numpy(view_mats).to(dev)
    mask_sum = torch.empty(H, W, device=dev, dtype=torch.int32)
    depth_sum = torch.empty(H, W, device=dev, dtype=torch.float32)
    vm = torch.empty(NS, H, W, device=dev, dtype=torch.uint8) if per_view else None
    vd, xs, ys = (torch.empty(NS, H, W, device=dev, dtype=torch.float32) for _ in range(3)) if per_view else (None,) * 3

    def ptr(t):
        return None if t is None else t.data_ptr()
    rc = _lib.load().mvster_geo_filter(ptr(dref), ptr(dsrc), ptr(ref_mats), ptr(view_mats), ptr(mask_sum), ptr(depth_sum),
                                       ptr(vm), ptr(vd), ptr(xs), ptr(ys), NS, H, W, float(pix_thres), float(rel_thres),
                                       torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc, "geo_filter")
    out = {"mask_sum": mask_sum, "depth_sum": depth_sum, "depth_ref": dref}
    if per_view:
        out.update(view_mask=vm.bool(), view_depth=vd, x_src=xs, y_src=ys)
    return out


def check_geometric_consistency(depth_ref, intrinsics_ref, extrinsics_ref, depth_src, intrinsics_src, extrinsics_src):
    """Reference signature (test_mvs4.py:313): -> (mask, depth_reprojected, x2d_src, y2d_src), NumPy in / NumPy out."""
    r = geometric_filter(depth_ref, intrinsics_ref, extrinsics_ref, [depth_src], [intrinsics_src], [extrinsics_src],
                         per_view=True)
    res = (r["view_mask"][0], r["view_depth"][0], r["x_src"][0], r["y_src"][0])
    if isinstance(depth_ref, torch.Tensor):
        return res
    return tuple(t.cpu().numpy() for t in res)


def filter_reference_view(ref_depth, ref_K, ref_E, confidence, src_depths, src_Ks, src_Es, conf_thres, thres_view,
                          ref_img=None):
    """The per-reference-view part of the reference's ``filter_depth`` (test_mvs4.py:352-407): photometric, geometric
    and final masks, the averaged depth (float64, like NumPy's float32 / int32 division) and the fused world points
    [M,3] of the pixels that pass; with ``ref_img`` [H,W,3] (float, 0..1, as ``read_img`` returns it) also their
    colours [M,3] uint8 (:395-396, :407).  Tensors stay on the GPU."""
    r = geometric_filter(ref_depth, ref_K, ref_E, src_depths, src_Ks, src_Es)
    dev = r["mask_sum"].device
    conf = _dev_depth(confidence, dev)
    photo_mask = conf > conf_thres
    geo_mask = r["mask_sum"] >= thres_view
    final_mask = photo_mask & geo_mask
    avg = (r["depth_sum"] + r["depth_ref"]).double() / (r["mask_sum"] + 1).double()
    H, W = avg.shape
    ys, xs = torch.nonzero(final_mask, as_tuple=True)                      # row-major order = NumPy's x[mask]
    depth = avg[ys, xs]
    kinv = torch.from_numpy(np.linalg.inv(_np32(ref_K)).astype(np.float64)).to(dev)
    einv = torch.from_numpy(np.linalg.inv(_np32(ref_E)).astype(np.float64)).to(dev)
    pix = torch.stack([xs.double() * depth, ys.double() * depth, depth])   # (x, y, 1) * depth
    cam = kinv @ pix
    world = (einv @ torch.cat([cam, torch.ones_like(depth)[None]], 0))[:3]
    out = dict(photo_mask=photo_mask, geo_mask=geo_mask, final_mask=final_mask, geo_mask_sum=r["mask_sum"],
               depth_est_averaged=avg, points=world.t().contiguous())
    if ref_img is not None:
        img = ref_img if isinstance(ref_img, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(ref_img))
        img = img.to(dev)
        if tuple(img.shape[:2]) != (H, W) or img.shape[-1] != 3:
            raise RuntimeError("filter_reference_view: ref_img must be [H,W,3] at the depth map's resolution")
        out["colors"] = (img[ys, xs] * 255).to(torch.uint8)              # (color * 255).astype(np.uint8): truncation
    return out


PLY_VERTEX_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1")])


def fuse_views(per_view_results):
    """Concatenate the ``points`` / ``colors`` of consecutive ``filter_reference_view`` results in reference-view order
    into the structured vertex array the reference hands to plyfile (test_mvs4.py:409-418): x, y, z float32 (the float64
    world points rounded once) and red, green, blue uint8."""
    pts = torch.cat([r["points"] for r in per_view_results], 0).to(torch.float32).cpu().numpy()
    cols = torch.cat([r["colors"] for r in per_view_results], 0).cpu().numpy()
    v = np.empty(len(pts), dtype=PLY_VERTEX_DTYPE)
    v["x"], v["y"], v["z"] = pts[:, 0], pts[:, 1], pts[:, 2]
    v["red"], v["green"], v["blue"] = cols[:, 0], cols[:, 1], cols[:, 2]
    return v


def write_ply(filename, vertices):
    """Binary little-endian PLY with one ``vertex`` element, the file ``PlyData([PlyElement.describe(v, 'vertex')])
    .write(f)`` produces for the array above (test_mvs4.py:420-421)."""
    v = np.ascontiguousarray(vertices, dtype=PLY_VERTEX_DTYPE)
    header = ("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\n"
              "property float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n" % len(v))
    with open(filename, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(v.tobytes())


def read_ply(filename):
    """Inverse of ``write_ply`` (round-trip tests and downstream checks)."""
    with open(filename, "rb") as f:
        data = f.read()
    end = data.index(b"end_header\n") + len(b"end_header\n")
    head = data[:end].decode("ascii").split("\n")
    n = int([l for l in head if l.startswith("element vertex")][0].split()[-1])
    if "format binary_little_endian 1.0" not in head:
        raise RuntimeError("read_ply: only the binary little-endian layout of write_ply is read")
    return np.frombuffer(data[end:end + n * PLY_VERTEX_DTYPE.itemsize], dtype=PLY_VERTEX_DTYPE).copy()
