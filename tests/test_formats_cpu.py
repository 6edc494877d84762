"""PFM / cam / pair readers and writers and the evaluation sample (mvster_amd/formats.py): round trips, hand-built files
and -- second half -- byte / element equality with what the reference's own functions write and read (fixture G10)."""
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


# ---- pinned against the reference's own readers / writers (fixture G10, oracle/make_golden.py g10) -------------------
import os

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "g10_formats.npz"))
PFM_CASES = ("grey", "grey1", "color", "big", "special")


def _same_bits(a, b):
    a, b = np.ascontiguousarray(a, dtype=np.float32), np.ascontiguousarray(b, dtype=np.float32)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("name", PFM_CASES)
def test_save_pfm_writes_the_reference_bytes(tmp_path, name):
    img = GOLD["pfm_%s_image" % name]
    if bool(GOLD["pfm_%s_image_big_endian" % name]):
        img = img.astype(">f4")
    p = tmp_path / "x.pfm"
    IO.save_pfm(str(p), img, float(GOLD["pfm_%s_scale_in" % name]))
    assert p.read_bytes() == GOLD["pfm_%s_bytes" % name].tobytes()


@pytest.mark.parametrize("name", PFM_CASES + ("hand",))
def test_read_pfm_reads_what_the_reference_reads(tmp_path, name):
    p = tmp_path / "x.pfm"
    p.write_bytes(GOLD["pfm_%s_bytes" % name].tobytes())
    data, scale = IO.read_pfm(str(p))
    assert scale == float(GOLD["pfm_%s_scale_out" % name])
    assert _same_bits(data, GOLD["pfm_%s_read" % name])                  # NaN / -0 / denormal payloads included
    if name != "hand":
        assert tuple(data.shape) == tuple(GOLD["pfm_%s_read_shape" % name])


def test_pfm_error_messages_are_the_reference_ones(tmp_path):
    for bad in ("magic", "dims"):
        p = tmp_path / (bad + ".pfm")
        p.write_bytes(GOLD["pfm_bad_%s_bytes" % bad].tobytes())
        with pytest.raises(Exception) as e:
            IO.read_pfm(str(p))
        assert str(e.value) == str(GOLD["pfm_bad_%s_message" % bad])
    for bad, img in (("dtype", np.zeros((2, 2), np.float64)), ("shape", np.zeros((2, 2, 2), np.float32))):
        with pytest.raises(Exception) as e:
            IO.save_pfm(str(tmp_path / "y.pfm"), img)
        assert str(e.value) == str(GOLD["pfm_bad_%s_message" % bad])


def _scan_dir(tmp_path):
    """Re-materialise the synthetic scan directory of the fixture (cam text as the reference's write_cam wrote it)."""
    H, W, nv = (int(x) for x in GOLD["ds_dims"])
    scan = "scan_g10"
    (tmp_path / scan / "cams").mkdir(parents=True)
    (tmp_path / scan / "images").mkdir()
    for v in range(nv):
        (tmp_path / scan / "cams" / ("%08d_cam.txt" % v)).write_bytes(GOLD["cam%d_text" % v].tobytes())
        (tmp_path / scan / "images" / ("%08d.jpg" % v)).write_bytes(GOLD["img%d_jpeg" % v].tobytes())
    (tmp_path / scan / "pair.txt").write_bytes(GOLD["pair_text"].tobytes())
    return scan, nv


def test_cam_and_pair_text_vs_reference(tmp_path):
    scan, nv = _scan_dir(tmp_path)
    for v in range(nv):
        p = tmp_path / "w.txt"
        IO.write_cam(str(p), GOLD["cam%d_array" % v])
        assert p.read_bytes() == GOLD["cam%d_text" % v].tobytes()
        K, E = IO.read_camera_parameters(str(tmp_path / scan / "cams" / ("%08d_cam.txt" % v)))
        assert K.dtype == np.float32 and E.dtype == np.float32
        assert np.array_equal(K, GOLD["cam%d_intrinsics" % v]) and np.array_equal(E, GOLD["cam%d_extrinsics" % v])
    pairs = IO.read_pair_file(str(tmp_path / scan / "pair.txt"))
    assert [r for r, _ in pairs] == GOLD["pair_refs"].tolist()
    assert [len(s) for _, s in pairs] == GOLD["pair_src_counts"].tolist()
    assert [x for _, s in pairs for x in s] == GOLD["pair_srcs_flat"].tolist()
    two = tmp_path / "two.txt"
    two.write_bytes(GOLD["cam_two_field_text"].tobytes())
    K4, _, dmin, itv = IO.read_cam_file(str(two), 1.06)
    assert [dmin, itv] == GOLD["cam_two_field_read"].tolist() and np.array_equal(K4, GOLD["cam_two_field_intrinsics"])


@pytest.mark.parametrize("tag,nviews,interval_scale", [("n3", 3, 1.06), ("n5", 5, 0.8)])
def test_eval_samples_vs_reference_dataset(tmp_path, tag, nviews, interval_scale):
    """What general_eval4.MVSDataset yields on the fixture's scan directory: view list, images, stage matrices, depth
    values, file name pattern -- element for element."""
    pytest.importorskip("PIL")
    scan, _ = _scan_dir(tmp_path)
    H, W, _ = (int(x) for x in GOLD["ds_dims"])
    metas = IO.eval_view_list(str(tmp_path), [scan], nviews)
    assert len(metas) == int(GOLD["ds_%s_len" % tag])
    assert [m[1] for m in metas] == GOLD["ds_%s_meta_ref" % tag].tolist()
    assert [len(m[2]) for m in metas] == GOLD["ds_%s_meta_src_counts" % tag].tolist()
    assert [x for m in metas for x in m[2]] == GOLD["ds_%s_meta_srcs_flat" % tag].tolist()
    for i, (sc, ref, srcs) in enumerate(metas):
        smp = IO.load_eval_sample(str(tmp_path), sc, ref, srcs, nviews, interval_scale=interval_scale, max_h=H, max_w=W)
        for st in ("stage1", "stage2", "stage3", "stage4"):
            want = GOLD["ds_%s_%d_%s" % (tag, i, st)]
            assert smp["proj_matrices"][st].dtype == want.dtype and np.array_equal(smp["proj_matrices"][st], want)
        dv = GOLD["ds_%s_%d_depth_values" % (tag, i)]
        assert smp["depth_values"].dtype == dv.dtype and np.array_equal(smp["depth_values"], dv)
        assert smp["filename"] == str(GOLD["ds_%s_%d_filename" % (tag, i)])
        if i == 0:
            assert np.array_equal(np.stack(smp["imgs"]), GOLD["ds_%s_0_imgs" % tag])


def test_eval_sample_refuses_to_resample(tmp_path):
    pytest.importorskip("PIL")
    scan, _ = _scan_dir(tmp_path)
    with pytest.raises(NotImplementedError):
        IO.load_eval_sample(str(tmp_path), scan, 0, [1, 2], 3, max_h=32, max_w=64)
