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


def reproject_with_depth(depth_ref, intrinsics_ref, extrinsics_ref, depth_src, intrinsics_src, extrinsics_src):
    """test_mvs4.py:273-310."""
    width, height = depth_ref.shape[1], depth_ref.shape[0]
    x_ref, y_ref = np.meshgrid(np.arange(0, width), np.arange(0, height))
    x_ref, y_ref = x_ref.reshape([-1]), y_ref.reshape([-1])
    xyz_ref = np.matmul(np.linalg.inv(intrinsics_ref), np.vstack((x_ref, y_ref, np.ones_like(x_ref))) * depth_ref.reshape([-1]))
    xyz_src = np.matmul(np.matmul(extrinsics_src, np.linalg.inv(extrinsics_ref)), np.vstack((xyz_ref, np.ones_like(x_ref))))[:3]
    K_xyz_src = np.matmul(intrinsics_src, xyz_src)
    xy_src = K_xyz_src[:2] / K_xyz_src[2:3]
    x_src = xy_src[0].reshape([height, width]).astype(np.float32)
    y_src = xy_src[1].reshape([height, width]).astype(np.float32)
    sampled_depth_src = remap_linear(depth_src, x_src, y_src)
    xyz_src = np.matmul(np.linalg.inv(intrinsics_src), np.vstack((xy_src, np.ones_like(x_ref))) * sampled_depth_src.reshape([-1]))
    xyz_reprojected = np.matmul(np.matmul(extrinsics_ref, np.linalg.inv(extrinsics_src)), np.vstack((xyz_src, np.ones_like(x_ref))))[:3]
    depth_reprojected = xyz_reprojected[2].reshape([height, width]).astype(np.float32)
    K_xyz_reprojected = np.matmul(intrinsics_ref, xyz_reprojected)
    with np.errstate(divide="ignore", invalid="ignore"):
        xy_reprojected = K_xyz_reprojected[:2] / K_xyz_reprojected[2:3]
    x_reprojected = xy_reprojected[0].reshape([height, width]).astype(np.float32)
    y_reprojected = xy_reprojected[1].reshape([height, width]).astype(np.float32)
    return depth_reprojected, x_reprojected, y_reprojected, x_src, y_src


def check_geometric_consistency(depth_ref, intrinsics_ref, extrinsics_ref, depth_src, intrinsics_src, extrinsics_src):
    """test_mvs4.py:313-328."""
    width, height = depth_ref.shape[1], depth_ref.shape[0]
    x_ref, y_ref = np.meshgrid(np.arange(0, width), np.arange(0, height))
    depth_reprojected, x2d_reprojected, y2d_reprojected, x2d_src, y2d_src = reproject_with_depth(
        depth_ref, intrinsics_ref, extrinsics_ref, depth_src, intrinsics_src, extrinsics_src)
    with np.errstate(divide="ignore", invalid="ignore"):
        dist = np.sqrt((x2d_reprojected - x_ref) ** 2 + (y2d_reprojected - y_ref) ** 2)
        depth_diff = np.abs(depth_reprojected - depth_ref)
        relative_depth_diff = depth_diff / depth_ref
        mask = np.logical_and(dist < 1, relative_depth_diff < 0.01)
    depth_reprojected[~mask] = 0
    return mask, depth_reprojected, x2d_src, y2d_src


def filter_reference_view(ref_depth, ref_K, ref_E, confidence, src_depths, src_Ks, src_Es, conf_thres, thres_view):
    """The per-reference-view part of filter_depth (test_mvs4.py:352-407): masks, averaged depth, world points."""
    photo_mask = confidence > conf_thres
    geo_mask_sum = 0
    reprojected, masks = [], []
    for d, K, E in zip(src_depths, src_Ks, src_Es):
        geo_mask, depth_reprojected, _, _ = check_geometric_consistency(ref_depth, ref_K, ref_E, d, K, E)
        geo_mask_sum = geo_mask_sum + geo_mask.astype(np.int32)
        reprojected.append(depth_reprojected)
        masks.append(geo_mask)
    depth_est_averaged = (sum(reprojected) + ref_depth) / (geo_mask_sum + 1)
    geo_mask = geo_mask_sum >= thres_view
    final_mask = np.logical_and(photo_mask, geo_mask)
    height, width = depth_est_averaged.shape[:2]
    x, y = np.meshgrid(np.arange(0, width), np.arange(0, height))
    x, y, depth = x[final_mask], y[final_mask], depth_est_averaged[final_mask]
    xyz_ref = np.matmul(np.linalg.inv(ref_K), np.vstack((x, y, np.ones_like(x))) * depth)
    xyz_world = np.matmul(np.linalg.inv(ref_E), np.vstack((xyz_ref, np.ones_like(x))))[:3]
    return dict(photo_mask=photo_mask, geo_mask=geo_mask, final_mask=final_mask, geo_mask_sum=geo_mask_sum,
                depth_est_averaged=depth_est_averaged, view_masks=masks, view_depths=reprojected,
                points=xyz_world.transpose((1, 0)))
