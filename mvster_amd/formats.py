"""On-disk formats either side of the path: PFM depth / confidence maps, MVSNet-style
``*_cam.txt`` and ``pair.txt``, the evaluation sample (view list, images, multi-scale ``proj_matrices`` dict,
``depth_values``) the forward pass takes.

Behaviour follows the reference's readers and writers -- ``datasets/data_io.py:6-71`` (``read_pfm`` / ``save_pfm``),
``datasets/general_eval4.py:24-57`` (view list), ``:59-79`` (``read_cam_file``), ``:81-86`` (``read_img``), ``:92-108``
(admissible image size), ``:111-188`` (sample), ``test_mvs4.py:94-103`` (``read_camera_parameters``), ``:126-136``
(``read_pair_file``), ``:138-155`` (``write_cam``).  **Pinned** by ``tests/golden/g10_formats.npz``: the bytes / text /
arrays the reference's own functions write and read for the same inputs (``oracle/make_golden.py g10``), compared
byte for byte and element for element in ``tests/test_formats_cpu.py``.
"""
import os
import re
import sys

import numpy as np

_PFM_DIMS = re.compile(r"^(\d+)\s(\d+)\s$")       # "<width> <height>\n": exactly what the reference accepts


def _pfm_header(f):
    """-> (channels, width, height, scale, numpy byte-order character) of an open PFM file."""
    magic = f.readline().decode("utf-8").rstrip()
    if magic not in ("PF", "Pf"):
        raise Exception("Not a PFM file.")
    dims = _PFM_DIMS.match(f.readline().decode("utf-8"))
    if dims is None:
        raise Exception("Malformed PFM header.")
    scale = float(f.readline().rstrip())
    # the sign of the scale line is the byte-order flag: negative = little-endian samples
    return (3 if magic == "PF" else 1), int(dims.group(1)), int(dims.group(2)), abs(scale), ("<" if scale < 0 else ">")


def read_pfm(filename):
    """-> (data [H,W] or [H,W,3] in top-to-bottom row order, dtype as stored, scale)."""
    with open(filename, "rb") as f:
        channels, width, height, scale, order = _pfm_header(f)
        samples = np.fromfile(f, order + "f")
    shape = (height, width, 3) if channels == 3 else (height, width)
    return samples.reshape(shape)[::-1], scale       # rows are stored bottom-to-top


def save_pfm(filename, image, scale=1):
    """float32 [H,W], [H,W,1] (both 'Pf') or [H,W,3] ('PF'); samples go out in the array's own byte order."""
    if image.dtype.name != "float32":
        raise Exception("Image dtype must be float32.")
    grey = image.ndim == 2 or (image.ndim == 3 and image.shape[2] == 1)
    if not grey and not (image.ndim == 3 and image.shape[2] == 3):
        raise Exception("Image must have H x W x 3, H x W x 1 or H x W dimensions.")
    order = image.dtype.byteorder
    little = order == "<" or (order == "=" and sys.byteorder == "little")
    header = "%s\n%d %d\n%f\n" % ("Pf" if grey else "PF", image.shape[1], image.shape[0], -scale if little else scale)
    with open(filename, "wb") as f:
        f.write(header.encode("utf-8"))
        image[::-1].tofile(f)


def _cam_lines(filename):
    with open(filename) as f:
        return [line.rstrip() for line in f.readlines()]


def read_camera_parameters(filename):
    """``extrinsic`` 4x4 on lines 1-4, ``intrinsic`` 3x3 on lines 7-9 -> (intrinsics, extrinsics), float32."""
    lines = _cam_lines(filename)
    extrinsics = np.array(" ".join(lines[1:5]).split(), dtype=np.float32).reshape((4, 4))
    intrinsics = np.array(" ".join(lines[7:10]).split(), dtype=np.float32).reshape((3, 3))
    return intrinsics, extrinsics


def read_cam_file(filename, interval_scale=1.0, ndepths=192):
    """Evaluation-time reader: intrinsics of the quarter-resolution stage (rows 0-1 divided by 4), depth_min and the
    depth interval (rescaled to ``ndepths`` planes when the file carries a plane count, then times interval_scale)."""
    lines = _cam_lines(filename)
    intrinsics, extrinsics = read_camera_parameters(filename)
    intrinsics[:2, :] /= 4.0
    fields = lines[11].split()
    depth_min, depth_interval = float(fields[0]), float(fields[1])
    if len(fields) >= 3:
        depth_max = depth_min + int(float(fields[2])) * depth_interval
        depth_interval = (depth_max - depth_min) / ndepths
    return intrinsics, extrinsics, depth_min, depth_interval * interval_scale


def write_cam(filename, cam):
    """cam [2,4,4]: [0] extrinsic, [1][:3,:3] intrinsic, [1][3] = depth_min, interval, planes, depth_max."""
    with open(filename, "w") as f:
        f.write("extrinsic\n")
        for i in range(4):
            f.write("".join(str(cam[0][i][j]) + " " for j in range(4)) + "\n")
        f.write("\nintrinsic\n")
        for i in range(3):
            f.write("".join(str(cam[1][i][j]) + " " for j in range(3)) + "\n")
        f.write("\n" + " ".join(str(cam[1][3][j]) for j in range(4)) + "\n")


def read_pair_file(filename):
    """-> [(ref_view, [src_view, ...]), ...]; views without sources are dropped."""
    data = []
    with open(filename) as f:
        for _ in range(int(f.readline())):
            ref_view = int(f.readline().rstrip())
            src_views = [int(x) for x in f.readline().rstrip().split()[1::2]]
            if src_views:
                data.append((ref_view, src_views))
    return data


def stage_proj_matrices(intrinsics, extrinsics):
    """Per-view (quarter-resolution) intrinsics [N,3,3] + extrinsics [N,4,4] -> the forward pass's dict
    ``stage1..stage4`` of [N,2,4,4] float32 ([.,0] = extrinsic, [.,1,:3,:3] = intrinsic scaled 1/2, 1, 2, 4)."""
    n = len(intrinsics)
    base = np.zeros((n, 2, 4, 4), dtype=np.float32)
    base[:, 0] = np.asarray(extrinsics, dtype=np.float32)
    base[:, 1, :3, :3] = np.asarray(intrinsics, dtype=np.float32)
    out = {}
    for name, s in (("stage1", 0.5), ("stage2", 1.0), ("stage3", 2.0), ("stage4", 4.0)):
        m = base.copy()
        m[:, 1, :2, :] = base[:, 1, :2, :] * s
        out[name] = m
    return out


def depth_value_range(depth_min, depth_interval, ndepths=192):
    """The ``depth_values`` vector of a sample (general_eval4.py:168-170)."""
    return np.arange(depth_min, depth_interval * (ndepths - 0.5) + depth_min, depth_interval, dtype=np.float32)


def read_img(filename):
    """8-bit image file -> float32 [H,W,3] in 0..1 (general_eval4.py:81-86)."""
    from PIL import Image
    return np.array(Image.open(filename), dtype=np.float32) / 255.0


def eval_view_list(datapath, scans, nviews):
    """[(scan, ref_view, src_views), ...] over the scans' ``pair.txt`` (general_eval4.py:24-57): reference views
    without sources are dropped, a source list shorter than ``nviews`` is filled up with its first entry."""
    metas = []
    for scan in scans:
        for ref_view, src_views in read_pair_file(os.path.join(datapath, scan, "pair.txt")):
            if len(src_views) < nviews:
                src_views = src_views + [src_views[0]] * (nviews - len(src_views))
            metas.append((scan, ref_view, src_views))
    return metas


def _admissible_size(h, w, max_h, max_w, base=64):
    """The size the reference's loader brings an image to (general_eval4.py:92-100): shrunk to fit max_h x max_w,
    then rounded down to multiples of ``base`` (float arithmetic as there)."""
    if h > max_h or w > max_w:
        scale = 1.0 * max_h / h
        if scale * w > max_w:
            scale = 1.0 * max_w / w
        return scale * w // base * base, scale * h // base * base
    return 1.0 * w // base * base, 1.0 * h // base * base


def load_eval_sample(datapath, scan, ref_view, src_views, nviews, interval_scale=1.06, ndepths=192,
                     max_h=None, max_w=None):
    """One evaluation sample as ``general_eval4.MVSDataset.__getitem__`` (:111-188) hands it to the forward pass:
    ``imgs`` list of [3,H,W] float32, ``proj_matrices`` dict stage1..4 of [N,2,4,4], ``depth_values`` [ndepths],
    ``filename`` pattern.  Images must already have an admissible size (H, W multiples of 64 within max_h x max_w):
    image resampling is outside the path and raises."""
    imgs, intr, extr, depth_values = [], [], [], None
    first_hw = None
    for i, vid in enumerate([ref_view] + list(src_views[:nviews - 1])):
        img_file = os.path.join(datapath, "{}/images_post/{:0>8}.jpg".format(scan, vid))
        if not os.path.exists(img_file):
            img_file = os.path.join(datapath, "{}/images/{:0>8}.jpg".format(scan, vid))
        img = read_img(img_file)
        K, E, depth_min, depth_interval = read_cam_file(
            os.path.join(datapath, "{}/cams/{:0>8}_cam.txt".format(scan, vid)), interval_scale, ndepths)
        h, w = img.shape[:2]
        new_w, new_h = _admissible_size(h, w, h if max_h is None else max_h, w if max_w is None else max_w)
        if (new_h, new_w) != (h, w) or (first_hw is not None and (h, w) != first_hw):
            raise NotImplementedError("image %s is %dx%d and would be resampled to %dx%d: resize the images first"
                                      % (img_file, h, w, *(first_hw or (new_h, new_w))))
        K[0, :] *= 1.0 * new_w / w
        K[1, :] *= 1.0 * new_h / h
        first_hw = first_hw or (h, w)
        imgs.append(img.transpose(2, 0, 1))
        intr.append(K)
        extr.append(E)
        if i == 0:
            depth_values = depth_value_range(depth_min, depth_interval, ndepths)
    return {"imgs": imgs, "proj_matrices": stage_proj_matrices(intr, extr), "depth_values": depth_values,
            "filename": scan + "/{}/" + "{:0>8}".format(ref_view) + "{}"}
