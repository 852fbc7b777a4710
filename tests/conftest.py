import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")
MEAN_PARAMS = os.path.join(REPO, "airpose_amd", "data", "smpl_mean_params.npz")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def rel_err(a, b):
    """max |a-b| / max |b|  (per-tensor normalised max error; the 1e-4 bar of north_star is on this)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.fixture(scope="session")
def golden():
    return {n: np.load(os.path.join(GOLDEN, n + ".npz"), allow_pickle=False)
            for n in ("copenet_b2", "hmr_b1", "geometry", "copenet_sep_b2", "singleview_b1", "muhmr_b1")}


@pytest.fixture(scope="session")
def copenet_sd(golden):
    from airpose_amd import weights as W
    return W.to_torch(W.copenet_state_dict(int(golden["copenet_b2"]["weights_seed"]), MEAN_PARAMS))


@pytest.fixture(scope="session")
def copenet_inputs(golden):
    import torch
    from airpose_amd import weights as W
    g = golden["copenet_b2"]
    return {k: torch.from_numpy(v) for k, v in W.synthetic_inputs(int(g["inputs_seed"]), int(g["batch"])).items()}


@pytest.fixture(scope="session")
def smplx_model():
    from airpose_amd import smplx_model
    return smplx_model.make_synthetic_model(4321)
