import os
import sys

import numpy as np
import pytest
import torch

# the tests' views are tiny: let them run the side-stream / split-update paths that the engine reserves for steps long
# enough to pay for them (c3 / c5 sizes), so that every test exercises what the full-size workloads run
os.environ.setdefault("STYLEMESH_OVERLAP_MIN_PIXELS", "0")

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, "tests", "golden")
for p in (REPO, os.path.join(REPO, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def batch_from_golden(d, prefix=""):
    """Rebuild the collated B = 1 batch tuple (view_contract.assemble_batch order) from a fixture."""
    t = lambda k: torch.from_numpy(d[prefix + k])
    uvs = []
    while prefix + f"uv{len(uvs)}" in d.files:
        uvs.append(t(f"uv{len(uvs)}"))
    eye = torch.eye(4, dtype=torch.float64)[None]
    return (t("rgb"), eye, eye.clone(), t("depth"), t("depth_level"), t("rounded_level"), t("other_level"),
            t("interp_weight"), torch.tensor([0]), uvs, t("mask"), t("angle_guidance"), t("angle_degrees"))


@pytest.fixture(scope="session")
def golden():
    return load_golden
