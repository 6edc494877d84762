"""PFM / cam / pair readers and writers (mvster_amd/formats.py): round trips and hand-built files."""
import struct

import numpy as np
import pytest

from mvster_amd import formats as IO


def test_pfm_round_trip_and_layout(tmp_path):
    img = np.arange(12, dtype=np.float32).reshape(3, 4) * 0.5
    p = tmp_path / "d.pfm"
    IO.save_pfm(str(p), img)
    raw = p.read_bytes()
    assert raw.startswith(b"Pf\n4 3\n-1.000000\n")                       # little-endian marker, width height
    body = raw[len(b"Pf\n4 3\n-1.000000\n"):]
    assert struct.unpack("<4f", body[:16]) == tuple(img[2])               # rows are stored bottom-to-top
    back, scale = IO.read_pfm(str(p))
    assert scale == 1.0 and back.dtype == np.float32 and np.array_equal(back, img)
    rgb = np.random.RandomState(0).rand(5, 6, 3).astype(np.float32)
    IO.save_pfm(str(p), rgb, scale=2)
    back, scale = IO.read_pfm(str(p))
    assert scale == 2.0 and np.array_equal(back, rgb)
    with pytest.raises(Exception):
        IO.save_pfm(str(p), img.astype(np.float64))


def test_pfm_big_endian_file(tmp_path):
    p = tmp_path / "be.pfm"
    vals = np.array([[1.5, -2.0], [3.25, 4.0]], dtype=">f4")
    p.write_bytes(b"Pf\n2 2\n1.0\n" + vals.tobytes())
    back, scale = IO.read_pfm(str(p))
    assert scale == 1.0 and np.array_equal(back, np.flipud(vals.astype(np.float32)))
    p.write_bytes(b"P6\n2 2\n1.0\n")
    with pytest.raises(Exception):
        IO.read_pfm(str(p))


def test_cam_and_pair_files(tmp_path):
    cam = np.zeros((2, 4, 4), dtype=np.float32)
    cam[0] = np.eye(4) + 0.01 * np.arange(16).reshape(4, 4)
    cam[1, :3, :3] = [[2892.33, 0, 823.2], [0, 2883.18, 619.07], [0, 0, 1]]
    cam[1, 3] = [425.0, 2.5, 192, 905.0]
    p = tmp_path / "00000000_cam.txt"
    IO.write_cam(str(p), cam)
    K, E = IO.read_camera_parameters(str(p))
    assert np.array_equal(K, cam[1, :3, :3]) and np.array_equal(E, cam[0])
    K4, E4, dmin, itv = IO.read_cam_file(str(p), interval_scale=1.06)
    assert np.allclose(K4[:2], cam[1, :2, :3] / 4) and K4[2, 2] == 1 and dmin == 425.0
    assert abs(itv - (425.0 + 192 * 2.5 - 425.0) / 192 * 1.06) < 1e-9
    dv = IO.depth_value_range(dmin, itv)
    assert dv.shape == (192,) and dv[0] == np.float32(425.0) and abs(dv[1] - dv[0] - itv) < 1e-3
    pm = IO.stage_proj_matrices([K4, K4], [E4, E4])
    assert pm["stage2"].shape == (2, 2, 4, 4) and np.array_equal(pm["stage4"][0, 1, :2, :3], K4[:2] * 4)
    assert np.array_equal(pm["stage1"][1, 0], E4) and pm["stage3"][0, 1, 2, 2] == 1
    q = tmp_path / "pair.txt"
    q.write_text("3\n0\n2 1 0.9 2 0.8\n1\n0\n2\n3 0 1.0 1 0.5 7 0.1\n")
    assert IO.read_pair_file(str(q)) == [(0, [1, 2]), (2, [0, 1, 7])]
