"""On-disk formats either side of the path: PFM depth / confidence maps, MVSNet-style
``*_cam.txt`` and ``pair.txt``, and the multi-scale ``proj_matrices`` dict the forward pass takes.

Behaviour follows the reference's readers and writers -- ``datasets/data_io.py:6-71`` (``read_pfm`` / ``save_pfm``),
``datasets/general_eval4.py:59-79`` (``read_cam_file``) and ``:173-188`` (stage matrices), ``test_mvs4.py:94-103``
(``read_camera_parameters``), ``:126-136`` (``read_pair_file``), ``:138-155`` (``write_cam``).  Those modules import
cv2 at load time and cannot be imported in this image, so these are restatements checked by round trips and
hand-built files (tests/test_formats_cpu.py), not by reference-generated goldens.
"""
import re
import sys

import numpy as np


def read_pfm(filename):
    """-> (data [H,W] or [H,W,3] float32 in top-to-bottom row order, scale)."""
    with open(filename, "rb") as f:
        header = f.readline().decode("utf-8").rstrip()
        if header == "PF":
            color = True
        elif header == "Pf":
            color = False
        else:
            raise Exception("Not a PFM file.")
        dims = re.match(r"^(\d+)\s(\d+)\s$", f.readline().decode("utf-8"))
        if not dims:
            raise Exception("Malformed PFM header.")
        width, height = map(int, dims.groups())
        scale = float(f.readline().rstrip())
        endian = "<" if scale < 0 else ">"           # a negative scale marks little-endian data
        scale = abs(scale)
        data = np.fromfile(f, endian + "f")
    shape = (height, width, 3) if color else (height, width)
    return np.flipud(np.reshape(data, shape)), scale    # PFM stores rows bottom-to-top


def save_pfm(filename, image, scale=1):
    image = np.flipud(image)
    if image.dtype.name != "float32":
        raise Exception("Image dtype must be float32.")
    if len(image.shape) == 3 and image.shape[2] == 3:
        color = True
    elif len(image.shape) == 2 or (len(image.shape) == 3 and image.shape[2] == 1):
        color = False
    else:
        raise Exception("Image must have H x W x 3, H x W x 1 or H x W dimensions.")
    endian = image.dtype.byteorder
    if endian == "<" or (endian == "=" and sys.byteorder == "little"):
        scale = -scale
    with open(filename, "wb") as f:
        f.write(("PF\n" if color else "Pf\n").encode("utf-8"))
        f.write("{} {}\n".format(image.shape[1], image.shape[0]).encode("utf-8"))
        f.write(("%f\n" % scale).encode("utf-8"))
        image.tofile(f)


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
