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
    """max |a-b| / max |b| over ONE semantic tensor (all entries of the same physical kind and scale).
    Do not pass a pred_pose (B,135) whole: its un-scaled translation (z ~ 10) would set the denominator for the
    O(1) 6-D rotation entries -- use pose_rel_errs, which normalises translation and rotation separately."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def pose_rel_errs(a, b):
    """rel_err of a pose vector (..,135) per semantic slice: translation [:3] and the 132 6-D rotation entries [3:]."""
    a, b = np.asarray(a), np.asarray(b)
    return {"trans": rel_err(a[..., :3], b[..., :3]), "rot6d": rel_err(a[..., 3:], b[..., 3:])}


def elem_err(a, b, atol):
    """element-wise max |a-b| / (atol + |b|): the like-for-like form of a '1e-4 relative' bar (atol = the scale below
    which an entry counts as zero, e.g. 1e-2 for O(1) rotation entries)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float((np.abs(a - b) / (atol + np.abs(b))).max())


def key_errs(name, a, b):
    """per-slice errors of one output-dict entry: pose tensors are split, everything else is one slice"""
    if name.startswith("pred_pose") or name.startswith("pose"):
        return {name + "." + k: v for k, v in pose_rel_errs(a, b).items()}
    return {name: rel_err(a, b)}


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
