"""Known-answer tests for the SMPL-X oracle (the only pin available: the upstream submodule and
model files are absent from the reference checkout — PARITY UNPINNED).  CPU only."""
import numpy as np
import torch

from airpose_amd import smplx_model as SM
from oracle import smplx_ref


def _eye(B, n):
    return torch.eye(3).expand(B, n, 3, 3).clone()


def _rot(axis, ang):
    axis = np.asarray(axis, np.float64)
    axis /= np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return torch.from_numpy((np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K).astype(np.float32))


def test_shapes_and_joint_count(smplx_model):
    B = 2
    v, j = smplx_ref.smplx_forward(smplx_model, torch.zeros(B, 10), _eye(B, 21))
    assert v.shape == (B, 10475, 3) and j.shape == (B, 127, 3)
    assert SM.NUM_OUT_JOINTS == 127
    assert (np.count_nonzero(smplx_model["lbs_weights"], axis=1) <= 4).all()


def test_identity_pose_zero_shape_is_template(smplx_model):
    v, j = smplx_ref.smplx_forward(smplx_model, torch.zeros(1, 10), _eye(1, 21))
    vt = smplx_model["v_template"]
    assert np.allclose(v[0].numpy(), vt, atol=2e-6)
    assert np.allclose(j[0, :55].numpy(), smplx_model["J_regressor"] @ vt, atol=2e-6)
    assert np.allclose(j[0, 55:76].numpy(), vt[smplx_model["extra_joint_verts"]], atol=2e-6)
    tri = smplx_model["faces"][smplx_model["lmk_faces_idx"]]
    lm = (vt[tri] * smplx_model["lmk_bary_coords"][:, :, None]).sum(1)
    assert np.allclose(j[0, 76:].numpy(), lm, atol=2e-6)


def test_identity_pose_shape_closed_form(smplx_model):
    betas = torch.randn(2, 10)
    v, j = smplx_ref.smplx_forward(smplx_model, betas, _eye(2, 21))
    want = smplx_model["v_template"][None] + np.einsum("bl,mkl->bmk", betas.numpy(),
                                                       smplx_model["shapedirs"][:, :, :10])
    assert np.allclose(v.numpy(), want, atol=5e-6)
    assert np.allclose(j[:, :55].numpy(), np.einsum("ji,bik->bjk", smplx_model["J_regressor"], want), atol=5e-6)


def test_single_joint_rotation_is_rigid_about_that_joint(smplx_model):
    """Rotate the left knee (joint 4): vertices bound with weight 1 to joints in its sub-tree rotate
    rigidly about the knee's rest position; vertices bound only outside the sub-tree stay put
    (posedirs zeroed so the pose corrective does not enter)."""
    m = dict(smplx_model)
    m["posedirs"] = np.zeros_like(m["posedirs"])
    R = _rot([0.3, 1.0, -0.2], 0.7)
    body = _eye(1, 21)
    body[0, 3] = R                                    # body_pose index 3 == joint 4
    v, j = smplx_ref.smplx_forward(m, torch.zeros(1, 10), body)
    vt = m["v_template"]
    J0 = m["J_regressor"] @ vt
    sub = {4}
    for k in range(55):
        if m["parents"][k] in sub:
            sub.add(k)
    W = m["lbs_weights"]
    insub = W[:, sorted(sub)].sum(1)
    full = np.nonzero(np.isclose(insub, 1.0, atol=1e-6))[0]
    none = np.nonzero(insub == 0)[0]
    assert len(full) > 50 and len(none) > 50
    want = (vt[full] - J0[4]) @ R.numpy().T + J0[4]
    assert np.allclose(v[0, full].numpy(), want, atol=5e-6)
    assert np.allclose(v[0, none].numpy(), vt[none], atol=5e-6)
    # posed joints: descendants rotate about the knee, the knee itself does not move
    assert np.allclose(j[0, 4].numpy(), J0[4], atol=5e-6)
    assert np.allclose(j[0, 7].numpy(), (J0[7] - J0[4]) @ R.numpy().T + J0[4], atol=5e-6)


def test_global_orient_rotates_about_pelvis_and_transl_adds(smplx_model):
    R = _rot([0, 0, 1], 1.1)
    t = torch.tensor([[0.3, -0.2, 5.0]])
    m = dict(smplx_model)
    m["posedirs"] = np.zeros_like(m["posedirs"])
    v, j = smplx_ref.smplx_forward(m, torch.zeros(1, 10), _eye(1, 21), global_orient=R[None, None], transl=t)
    vt = m["v_template"]
    J0 = m["J_regressor"] @ vt
    want = (vt - J0[0]) @ R.numpy().T + J0[0] + t.numpy()
    assert np.allclose(v[0].numpy(), want, atol=1e-5)


def test_sparse_skin_weights_roundtrip(smplx_model):
    idx, w = SM.sparse_skin_weights(smplx_model["lbs_weights"])
    assert idx.shape[1] == 4
    dense = np.zeros_like(smplx_model["lbs_weights"])
    np.add.at(dense, (np.arange(dense.shape[0])[:, None], idx), w)
    assert np.array_equal(dense, smplx_model["lbs_weights"])
    nz = w != 0
    assert (np.diff(np.where(nz, idx, 1000), axis=1) >= 0).all()    # ascending bone order, padding last
