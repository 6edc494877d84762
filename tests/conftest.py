import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun / the driver)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


class Golden:
    """npz fixture with tensors on demand."""

    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name + ".npz"))

    def keys(self):
        return list(self.z.keys())

    def np(self, k):
        return self.z[k]

    def t(self, k, device="cpu"):
        return torch.from_numpy(np.ascontiguousarray(self.z[k])).to(device)

    def __contains__(self, k):
        return k in self.z


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]
    return get


SHIPPED = dict(arch_mode="fpn", reg_net="reg2d", num_stage=4, fpn_base_channel=8, reg_channel=8,
               stage_splits=[8, 8, 4, 4], depth_interals_ratio=[0.5, 0.5, 0.5, 1], group_cor=True,
               group_cor_dim=[8, 8, 4, 4], inverse_depth=True, agg_type="ConvBnReLU3D", dcn=False, pos_enc=0,
               mono=True, asff=False, attn_temp=2, attn_fuse_d=True)


@pytest.fixture(scope="session")
def shipped_cfg():
    return dict(SHIPPED)


@pytest.fixture(scope="session")
def checkpoint(golden):
    g = golden("g7_checkpoint")
    return {k: g.t(k) for k in g.keys()}
