"""The policy of ``graph.ForwardCache`` (the transparent hipGraph cache inside ``MVS4net.forward``) without a GPU: what is
eager, what is captured, what invalidates an entry, what the key separates.  The capture itself (``_CachedForward``) is
replaced by a recorder here; the real one is exercised by tests/test_gpu_model.py on the GPU."""
import torch

from mvster_amd import graph


class _FakeModel:
    """Just what ForwardCache reads of MVS4net."""
    warp_variant, overlap_streams, attn_temp, attn_fuse_d, num_stage = 0, True, 2, True, 4
    stage_splits, depth_interals_ratio, group_cor_dim = [8, 8, 4, 4], [0.5, 0.5, 0.5, 1], [8, 8, 4, 4]

    def __init__(self):
        self.stamp = 1
        self.eager_calls = 0

    def _state_stamp(self):
        return self.stamp

    def _forward_eval(self, imgs, proj, dv):
        self.eager_calls += 1
        return {"depth": imgs[0].sum() + dv.sum(), "how": "eager"}


class _FakeCaptured:
    built = 0

    def __init__(self, model, imgs, proj, dv):
        type(self).built += 1
        self.loaded = (imgs, proj, dv)

    def load(self, imgs, proj, dv):
        self.loaded = (imgs, proj, dv)

    def replay(self):
        imgs, _, dv = self.loaded
        return {"depth": imgs[0].sum() + dv.sum(), "how": "replay"}


def _sample(h=64, w=64, n=3, fill=1.0):
    imgs = [torch.full((1, 3, h, w), fill) for _ in range(n)]
    proj = {"stage%d" % s: torch.zeros(1, n, 2, 4, 4) for s in range(1, 5)}
    return imgs, proj, torch.tensor([[425.0, 935.0]])


def _cache(monkeypatch, capacity=4):
    monkeypatch.setattr(graph, "_CachedForward", _FakeCaptured)
    _FakeCaptured.built = 0
    return graph.ForwardCache(capacity=capacity)


def test_first_call_eager_second_captured_then_replayed(monkeypatch):
    c, m = _cache(monkeypatch), _FakeModel()
    a = c(m, *_sample())
    assert a["how"] == "eager" and m.eager_calls == 1 and _FakeCaptured.built == 0
    b = c(m, *_sample(fill=2.0))                       # same shape, other values: capture + replay on THESE inputs
    assert b["how"] == "replay" and _FakeCaptured.built == 1 and float(b["depth"]) == 2.0 * 3 * 64 * 64 + 1360
    d = c(m, *_sample(fill=3.0))
    assert d["how"] == "replay" and _FakeCaptured.built == 1 and float(d["depth"]) == 3.0 * 3 * 64 * 64 + 1360
    assert c.stats == {"eager": 1, "captured": 1, "replayed": 2, "capture_failed": 0}


def test_parameter_change_goes_back_to_eager_then_recaptures(monkeypatch):
    c, m = _cache(monkeypatch), _FakeModel()
    c(m, *_sample())
    c(m, *_sample())
    m.stamp = 2                                          # an optimizer step / EMA swap / load_state_dict
    assert c(m, *_sample())["how"] == "eager" and m.eager_calls == 2
    assert c(m, *_sample())["how"] == "replay" and _FakeCaptured.built == 2
    m.stamp = 3                                          # weights that change at EVERY call are never captured
    for k in range(4):
        m.stamp += 1
        assert c(m, *_sample())["how"] == "eager"
    assert _FakeCaptured.built == 2


def test_key_separates_shapes_views_and_launch_attributes(monkeypatch):
    c, m = _cache(monkeypatch), _FakeModel()
    for args in (_sample(64, 64, 3), _sample(64, 128, 3), _sample(64, 64, 5)):
        assert c(m, *args)["how"] == "eager"
    assert len(c.entries) == 3
    for args in (_sample(64, 64, 3), _sample(64, 128, 3), _sample(64, 64, 5)):
        assert c(m, *args)["how"] == "replay"
    m.warp_variant = 3                                   # another kernel form = another launch sequence
    assert c(m, *_sample(64, 64, 3))["how"] == "eager"
    m.warp_variant = 0
    assert c(m, *_sample(64, 64, 3))["how"] == "replay"
    wide = _sample(64, 64, 3)
    wide = (wide[0], wide[1], torch.tensor([[425.0, 600.0, 935.0]]))     # another depth_values width
    assert c(m, *wide)["how"] == "eager"


def test_least_recently_used_entry_is_dropped(monkeypatch):
    c, m = _cache(monkeypatch, capacity=2), _FakeModel()
    shapes = [(64, 64), (64, 128), (128, 128)]
    for h, w in shapes[:2]:
        c(m, *_sample(h, w))
        c(m, *_sample(h, w))
    c(m, *_sample(*shapes[0]))                           # touch the first: the second is now the oldest
    c(m, *_sample(*shapes[2]))                           # third shape: evicts the second
    assert len(c.entries) == 2
    assert c(m, *_sample(*shapes[0]))["how"] == "replay"
    assert c(m, *_sample(*shapes[1]))["how"] == "eager"


def test_failed_capture_keeps_the_shape_on_eager_launches(monkeypatch):
    class Failing(_FakeCaptured):
        def __init__(self, *a):
            raise RuntimeError("operation not permitted when stream is capturing")
    c, m = _cache(monkeypatch), _FakeModel()
    monkeypatch.setattr(graph, "_CachedForward", Failing)
    c(m, *_sample())
    import warnings
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert c(m, *_sample())["how"] == "eager"
    assert len(w) == 1 and "stays on eager launches" in str(w[0].message)
    assert c(m, *_sample())["how"] == "eager" and c.stats["capture_failed"] == 1


def test_dense_views():
    t = torch.arange(24.0).view(2, 3, 4)
    assert graph._is_dense(t) and graph._is_dense(t.permute(0, 2, 1)) and graph._is_dense(t[1]) and graph._is_dense(t[:1])
    assert not graph._is_dense(t[:, :2]) and not graph._is_dense(t[:, :, ::2]) and not graph._is_dense(t[:, :, :1])
    assert graph._is_dense(torch.zeros(1, 5, 1, 7).permute(3, 0, 2, 1))
