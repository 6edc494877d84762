"""The geometric-filter restatement (oracle/geo_filter_oracle.py) on an analytic scene: properties that must hold
whatever OpenCV build the reference ran on (its own functions cannot be imported here: cv2 is absent)."""
import numpy as np

from mvster_amd.synthetic_scene import plane_depth_maps
from oracle import geo_filter_oracle as GO


def test_remap_linear_quantises_like_opencv():
    src = np.arange(20, dtype=np.float32).reshape(4, 5)       # src[y][x] = 5*y + x
    mx = np.array([[0.0, 1.5, 3.99, -0.5, 4.25, 1.0 + 1 / 64 + 1e-4, 1.0 + 1 / 64 - 1e-4]], dtype=np.float32)
    my = np.array([[0.0, 0.0, 1.0, 1.0, 2.0, 0.0, 0.0]], dtype=np.float32)
    out = GO.remap_linear(src, mx, my)[0]
    assert out[0] == 0.0 and out[1] == 1.5                    # exact taps and midpoints survive the 1/32 grid
    assert out[2] == 9.0                                      # 3.99 * 32 = 127.68 -> 128: sampled AT x = 4
    assert out[3] == 2.5                                      # half outside on the left: the border value 0 weighs in
    assert out[4] == np.float32(14 * 0.75)                    # x = 4.25: the east tap is outside -> 0
    assert out[5] == np.float32(1 + 1 / 32)                   # just above 1/64 rounds up to the next 1/32 step ...
    assert out[6] == 1.0                                      # ... just below rounds down


def test_plane_scene_is_geometrically_consistent():
    depths, Ks, Es = plane_depth_maps(4, 48, 64, seed=1)
    res = GO.filter_reference_view(depths[0], Ks[0], Es[0], np.ones_like(depths[0]), depths[1:], Ks[1:], Es[1:], 0.5, 3)
    vis = res["geo_mask_sum"] == 3
    assert vis.mean() > 0.5                                   # the plane is seen by all views in most of the image
    assert np.abs(res["depth_est_averaged"][vis] - depths[0][vis]).max() < 2e-2
    n = np.array([0.15, -0.1, 1.0]); n /= np.linalg.norm(n)
    assert np.abs(res["points"] @ n - 650.0).max() < 5e-2     # fused points lie on the plane
    # corrupt one source view: its pixels stop voting, the others keep theirs
    bad = depths.copy()
    bad[2] *= 1.05
    res2 = GO.filter_reference_view(bad[0], Ks[0], Es[0], np.ones_like(depths[0]), bad[1:], Ks[1:], Es[1:], 0.5, 3)
    assert res2["view_masks"][1].mean() < 0.02
    assert (res2["view_masks"][0] == res["view_masks"][0]).all()


def test_vertex_array_and_ply_round_trip(tmp_path):
    """Colours are truncated to uint8 in reference-view order (test_mvs4.py:395-396, :407-418); the PLY writer of the
    product stores exactly the structured array (host code, no GPU involved)."""
    from mvster_amd import fusion
    depths, Ks, Es = plane_depth_maps(3, 32, 48, seed=2)
    rng = np.random.RandomState(0)
    views = []
    for ref in (0, 1):
        order = [ref] + [v for v in range(3) if v != ref]
        img = rng.rand(32, 48, 3).astype(np.float32)
        r = GO.filter_reference_view(depths[order[0]], Ks[order[0]], Es[order[0]], np.ones((32, 48), np.float32),
                                     depths[order[1:]], [Ks[v] for v in order[1:]], [Es[v] for v in order[1:]], 0.5, 1,
                                     ref_img=img)
        assert r["colors"].dtype == np.uint8 and r["colors"].shape == (int(r["final_mask"].sum()), 3)
        assert np.array_equal(r["colors"], np.floor(img[r["final_mask"]] * 255).astype(np.uint8))
        views.append(r)
    v = GO.vertex_array(views)
    assert len(v) == sum(len(r["points"]) for r in views) and v.dtype.names == ("x", "y", "z", "red", "green", "blue")
    path = str(tmp_path / "scan.ply")
    fusion.write_ply(path, v)
    back = fusion.read_ply(path)
    assert back.dtype == fusion.PLY_VERTEX_DTYPE and np.array_equal(back["x"], v["x"]) and np.array_equal(back["blue"], v["blue"])
    head = open(path, "rb").read(200).decode("ascii", "ignore")
    assert head.startswith("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x" % len(v))
