"""AirPose+ fitting loop oracle (oracle/fitting_ref.py, parity unpinned: its third-party pieces are absent upstream) --
known-answer and self-consistency tests on CPU."""
import numpy as np
import torch

from oracle import fitting_ref as Fr
from oracle import geometry_ref


def test_decoder_rotations_and_axis_angle_round_trip(smplx_model):
    """VPoser.decode: the Gram-Schmidt output is a rotation, matrot2aa inverts lbs.batch_rodrigues on it."""
    vp, init, _, _ = Fr.synthetic_problem(smplx_model, L=6)
    aa, R = Fr.vposer_decode(vp, init["z"])
    I = torch.eye(3, dtype=R.dtype)
    assert (R.transpose(-1, -2) @ R - I).abs().max() < 1e-12 and (torch.linalg.det(R) - 1).abs().max() < 1e-12
    back = Fr.lbs_batch_rodrigues(aa.reshape(-1, 3)).view_as(R)
    assert (back - R).abs().max() < 1e-7
    # pytorch3d 6-D convention: rows; the identity vector gives the identity matrix
    assert torch.equal(Fr.rotation_6d_to_matrix(torch.tensor([[1.0, 0, 0, 0, 1.0, 0]])), torch.eye(3).unsqueeze(0))


def test_body_joints_rest_pose_and_shape(smplx_model):
    """BodyModel.Jtr: zero pose gives the regressed rest joints of the shaped template."""
    beta = torch.linspace(-1, 1, 10, dtype=torch.float64)
    J = Fr.body_joints(smplx_model, torch.zeros(2, 63, dtype=torch.float64), beta)
    vs = torch.as_tensor(smplx_model["v_template"], dtype=torch.float64) + \
        torch.einsum("l,mkl->mk", beta, torch.as_tensor(smplx_model["shapedirs"], dtype=torch.float64)[:, :, :10])
    want = torch.as_tensor(smplx_model["J_regressor"], dtype=torch.float64) @ vs
    assert (J[0] - want).abs().max() < 1e-6 and torch.equal(J[0], J[1])     # (rodrigues of 0 carries the 1e-8 offset)


def test_objective_gradients_by_finite_differences(smplx_model):
    vp, init, data, gt = Fr.synthetic_problem(smplx_model, L=8)
    total, grads, parts = Fr.loss_and_grads(vp, smplx_model, init, data, 3)
    assert float(parts["loss_2d"]) > 0 and float(parts["loss_temporal"]) > 0
    for k, idx in (("z", (2, 5)), ("phi0", (1, 4)), ("phi1", (7, 0)), ("tau0", (0, 2)), ("tau1", (3, 1)), ("beta", (6,))):
        eps = 1e-6
        p = {kk: vv.clone() for kk, vv in init.items()}
        p[k][idx] += eps
        lp, _ = Fr.loss_terms(vp, smplx_model, p, data, 3)
        p[k][idx] -= 2 * eps
        lm, _ = Fr.loss_terms(vp, smplx_model, p, data, 3)
        fd = float((lp - lm) / (2 * eps))
        assert abs(fd - float(grads[k][idx])) < 1e-6 * max(1.0, abs(fd)), k
    # hips lose half their weight on every iteration (the script divides the confidences in place)
    l0, _ = Fr.loss_terms(vp, smplx_model, init, data, 0)
    l9, _ = Fr.loss_terms(vp, smplx_model, init, data, 9)
    assert float(l9) < float(l0)


def test_fit_reduces_the_objective(smplx_model):
    vp, init, data, gt = Fr.synthetic_problem(smplx_model, L=8)
    res, losses = Fr.fit(vp, smplx_model, init, data, n_iters=30)
    assert losses[-1] < 0.8 * losses[0]
    assert all(torch.isfinite(v).all() for v in res.values())
