"""CPU restatement of the reference's geometric-consistency filter.  TEST INFRASTRUCTURE ONLY.

Follows /root/reference/test_mvs4.py:273-328 (``reproject_with_depth``, ``check_geometric_consistency``) and the
per-reference-view accumulation of ``filter_depth`` (:362-385, :397-407) in NumPy, with the reference's dtype flow
(float32 inverses, float64 geometry, float32 maps).

Parity status: UNPINNED.  ``test_mvs4.py`` imports cv2 / plyfile / tensorboardX, none of which exist in this image,
so the reference's own functions cannot be run here to produce golden vectors.  The one third-party piece on the
path is ``cv2.remap(depth_src, x_src, y_src, interpolation=cv2.INTER_LINEAR)`` (OpenCV, version not pinned by the
reference's requirements.txt): restated below from OpenCV's published algorithm -- float maps are converted to
fixed point with INTER_BITS = 5 (coordinates rounded to 1/32 pixel, round-half-even), the four taps are weighted
with the float bilinear table and taps outside the image take the constant border value 0.

Only ``tests/`` may import this module.
"""
import numpy as np

INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS


def remap_linear(src, map_x, map_y):
    """cv2.remap(src, map_x, map_y, INTER_LINEAR), borderMode=BORDER_CONSTANT, borderValue=0, float32 source."""
    H, W = src.shape
    sx = np.rint(map_x.astype(np.float64) * INTER_TAB_SIZE).astype(np.int64)       # cvRound: round half to even
    sy = np.rint(map_y.astype(np.float64) * INTER_TAB_SIZE).astype(np.int64)
    ix, iy = sx >> INTER_BITS, sy >> INTER_BITS
    fx = (sx & (INTER_TAB_SIZE - 1)).astype(np.float32) / np.float32(INTER_TAB_SIZE)
    fy = (sy & (INTER_TAB_SIZE - 1)).astype(np.float32) / np.float32(INTER_TAB_SIZE)

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        return np.where(ok, src[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)], np.float32(0)).astype(np.float32)
    one = np.float32(1)
    w00, w01 = (one - fy) * (one - fx), (one - fy) * fx
    w10, w11 = fy * (one - fx), fy * fx
    return (tap(iy, ix) * w00 + tap(iy, ix + 1) * w01 + tap(iy + 1, ix) * w10 + tap(iy + 1, ix + 1) * w11).astype(np.float32)


def _pixel_grid(height, width):
    """Homogeneous integer pixel coordinates [3, H*W] (x fastest), as np.meshgrid + vstack give them (int64)."""
    xs, ys = np.meshgrid(np.arange(0, width), np.arange(0, height))
    return np.vstack((xs.reshape(-1), ys.reshape(-1), np.ones(height * width, dtype=xs.dtype)))


def _lift(K, pix_h, depth_flat):
    """inv(K) (float32, like np.linalg.inv of a float32 matrix) times depth-scaled homogeneous pixels (float64)."""
    return np.matmul(np.linalg.inv(K), pix_h * depth_flat)


def _rigid(E_to, E_from, pts):
    """Points of camera `from` in camera `to`: (E_to inv(E_from)) in float32, applied to float64 homogeneous points."""
    rel = np.matmul(E_to, np.linalg.inv(E_from))
    return np.matmul(rel, np.vstack((pts, np.ones((1, pts.shape[1]), dtype=np.int64))))[:3]


def _perspective(K, pts):
    uvw = np.matmul(K, pts)
    with np.errstate(divide="ignore", invalid="ignore"):
        return uvw[:2] / uvw[2:3]


def reproject_with_depth(depth_ref, intrinsics_ref, extrinsics_ref, depth_src, intrinsics_src, extrinsics_src):
    """Round trip reference pixel -> source view -> back, with the source depth sampled in between.  Restates
    test_mvs4.py:273-310 step by step (lift with the reference depth :280-281, source camera :283-284, source
    pixels :286-287, remap :291-293, lift with the sampled depth :298-299, back :301-302, re-project :304-308);
    returns (depth_reprojected, x_reprojected, y_reprojected, x_src, y_src) as float32 maps."""
    height, width = depth_ref.shape
    grid = _pixel_grid(height, width)
    in_src_cam = _rigid(extrinsics_src, extrinsics_ref, _lift(intrinsics_ref, grid, depth_ref.reshape(-1)))
    xy_src = _perspective(intrinsics_src, in_src_cam)
    x_src = xy_src[0].reshape(height, width).astype(np.float32)
    y_src = xy_src[1].reshape(height, width).astype(np.float32)
    sampled = remap_linear(depth_src, x_src, y_src)
    src_h = np.vstack((xy_src, np.ones((1, height * width), dtype=np.int64)))
    back = _rigid(extrinsics_ref, extrinsics_src, _lift(intrinsics_src, src_h, sampled.reshape(-1)))
    xy_back = _perspective(intrinsics_ref, back)
    as_map = lambda v: v.reshape(height, width).astype(np.float32)     # noqa: E731
    return as_map(back[2]), as_map(xy_back[0]), as_map(xy_back[1]), x_src, y_src


def check_geometric_consistency(depth_ref, intrinsics_ref, extrinsics_ref, depth_src, intrinsics_src, extrinsics_src):
    """Vote of one source view (test_mvs4.py:313-328): a pixel is consistent if it comes back within 1 px (:319) and
    within 1 % relative depth (:322-323); inconsistent pixels get reprojected depth 0 (:326)."""
    height, width = depth_ref.shape
    depth_back, x_back, y_back, x_src, y_src = reproject_with_depth(depth_ref, intrinsics_ref, extrinsics_ref, depth_src,
                                                                   intrinsics_src, extrinsics_src)
    cols, rows = np.meshgrid(np.arange(0, width), np.arange(0, height))
    with np.errstate(divide="ignore", invalid="ignore"):
        pixel_error = np.sqrt((x_back - cols) ** 2 + (y_back - rows) ** 2)
        relative_error = np.abs(depth_back - depth_ref) / depth_ref
        mask = np.logical_and(pixel_error < 1, relative_error < 0.01)
    depth_back[~mask] = 0
    return mask, depth_back, x_src, y_src


def filter_reference_view(ref_depth, ref_K, ref_E, confidence, src_depths, src_Ks, src_Es, conf_thres, thres_view,
                          ref_img=None):
    """What filter_depth does for one reference view (test_mvs4.py:352-407): photometric mask from the confidence
    (:361), votes and reprojected depths of every source view (:369-383), their average with the reference depth
    (:385), the >= thres_view geometric mask (:387-388) and the surviving pixels lifted to world space (:399-407)."""
    photo_mask = confidence > conf_thres
    votes = np.zeros(ref_depth.shape, dtype=np.int32)
    depth_sum = 0
    view_masks, view_depths = [], []
    for d, K, E in zip(src_depths, src_Ks, src_Es):
        ok, depth_back, _, _ = check_geometric_consistency(ref_depth, ref_K, ref_E, d, K, E)
        votes = votes + ok.astype(np.int32)
        depth_sum = depth_sum + depth_back                  # python sum(): 0 + d1 + d2 + ...
        view_masks.append(ok)
        view_depths.append(depth_back)
    depth_est_averaged = (depth_sum + ref_depth) / (votes + 1)
    geo_mask = votes >= thres_view
    final_mask = np.logical_and(photo_mask, geo_mask)
    height, width = depth_est_averaged.shape[:2]
    cols, rows = np.meshgrid(np.arange(0, width), np.arange(0, height))
    sel_x, sel_y, sel_d = cols[final_mask], rows[final_mask], depth_est_averaged[final_mask]
    cam = np.matmul(np.linalg.inv(ref_K), np.vstack((sel_x, sel_y, np.ones_like(sel_x))) * sel_d)
    world = np.matmul(np.linalg.inv(ref_E), np.vstack((cam, np.ones_like(sel_x))))[:3]
    out = dict(photo_mask=photo_mask, geo_mask=geo_mask, final_mask=final_mask, geo_mask_sum=votes,
               depth_est_averaged=depth_est_averaged, view_masks=view_masks, view_depths=view_depths,
               points=world.transpose((1, 0)))
    if ref_img is not None:
        out["colors"] = (ref_img[final_mask] * 255).astype(np.uint8)          # test_mvs4.py:395-396, :407
    return out


def vertex_array(per_view):
    """The structured array filter_depth builds for plyfile from all reference views (test_mvs4.py:409-418)."""
    pts = np.concatenate([r["points"] for r in per_view], axis=0)
    cols = np.concatenate([r["colors"] for r in per_view], axis=0)
    xyz = np.array([tuple(v) for v in pts], dtype=[("x", "f4"), ("y", "f4"), ("z", "f4")])
    rgb = np.array([tuple(v) for v in cols], dtype=[("red", "u1"), ("green", "u1"), ("blue", "u1")])
    allv = np.empty(len(xyz), xyz.dtype.descr + rgb.dtype.descr)
    for name in xyz.dtype.names:
        allv[name] = xyz[name]
    for name in rgb.dtype.names:
        allv[name] = rgb[name]
    return allv
