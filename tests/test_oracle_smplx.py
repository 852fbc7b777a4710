"""Known-answer tests for the SMPL-X oracle (the only pin available: the upstream submodule and
model files are absent from the reference checkout — PARITY UNPINNED).  CPU only."""
import numpy as np
import pytest
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


# ------------------------------------------------------------------------------------------------ model-file loader
def _write_upstream_npz(path, md, wide):
    """An npz in the key layout of the distributed SMPLX_{GENDER}.npz (upstream smplx 0.1.28 reads exactly these keys):
    posedirs (V,3,486), shapedirs (V,3,400) = 300 shape + 100 expression  (or (V,3,20) = 10 + 10 for the small file),
    kintree_table (2,J) with the root's parent stored as a large unsigned value."""
    V = md["v_template"].shape[0]
    rs = np.random.RandomState(5)
    if wide:
        sd = (rs.standard_normal((V, 3, 400)) * 0.01).astype(np.float32)
        sd[:, :, :10] = md["shapedirs"][:, :, :10]
        sd[:, :, 300:310] = md["shapedirs"][:, :, 10:]
    else:
        sd = md["shapedirs"].copy()
    kt = np.stack([md["parents"].copy(), np.arange(55)]).astype(np.int64)
    kt[0, 0] = 4294967295
    np.savez(path, v_template=md["v_template"], f=md["faces"].astype(np.uint32), shapedirs=sd,
             posedirs=md["posedirs"].T.reshape(V, 3, 486).copy(), J_regressor=md["J_regressor"],
             kintree_table=kt, weights=md["lbs_weights"], lmk_faces_idx=md["lmk_faces_idx"].astype(np.int32),
             lmk_bary_coords=md["lmk_bary_coords"].astype(np.float64))


@pytest.mark.parametrize("wide", [True, False])
def test_load_model_npz_round_trip(tmp_path, wide):
    """load_model_npz is the only route to a real SMPL-X file: an npz written in the upstream key layout must come
    back as the dict it was made from (both the 400-column and the 10+10-column shapedirs files; in the small file
    the expression directions are columns 10:20, as upstream's SMPLX.__init__ takes them)."""
    from airpose_amd import smplx_model as SM
    from oracle import smplx_ref
    md = SM.make_synthetic_model(99, num_verts=64, num_faces=40, max_bones=6)
    p = str(tmp_path / "SMPLX_NEUTRAL.npz")
    _write_upstream_npz(p, md, wide)
    assert SM.find_model(str(tmp_path), "neutral") == p and SM.find_model(p) == p
    got = SM.load_model_npz(p)
    for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "parents", "lbs_weights", "faces", "lmk_faces_idx",
              "lmk_bary_coords"):
        assert got[k].shape == md[k].shape, k
        assert np.array_equal(got[k], md[k].astype(got[k].dtype)), k
    assert got["parents"][0] == -1 and got["synthetic"] is False
    # expression inputs must act through the loaded file exactly as through the source dict
    got["extra_joint_verts"] = md["extra_joint_verts"]        # (the 64-vertex toy has no vertex 9120)
    gen = torch.Generator().manual_seed(3)
    betas, expr = torch.randn(2, 10, generator=gen), torch.randn(2, 10, generator=gen)
    bp = torch.eye(3).expand(2, 21, 3, 3)
    v0, j0 = smplx_ref.smplx_forward(md, betas, bp, expression=expr)
    v1, j1 = smplx_ref.smplx_forward(got, betas, bp, expression=expr)
    assert torch.equal(v0, v1) and torch.equal(j0, j1)
    vz, _ = smplx_ref.smplx_forward(got, betas, bp)
    assert (v1 - vz).abs().max() > 1e-3                        # ... and they do act


def test_sparse_skin_weights_keep_every_bone():
    """K = max non-zeros per row (>= 4): rows with 5..9 bones must survive the top-K packing un-truncated."""
    from airpose_amd import smplx_model as SM
    for mb in (4, 6, 9):
        md = SM.make_synthetic_model(7, num_verts=300, num_faces=60, max_bones=mb)
        W = md["lbs_weights"]
        idx, w = SM.sparse_skin_weights(W)
        assert idx.shape[1] == max(mb, 4) and (W != 0).sum(1).max() == mb
        dense = np.zeros_like(W)
        for k in range(idx.shape[1]):
            np.add.at(dense, (np.arange(W.shape[0]), idx[:, k]), w[:, k])
        assert np.array_equal(dense, W)
        live = np.where(w != 0, idx, 10 ** 6)
        assert (np.diff(live, axis=1) >= 0).all()              # ascending bone order, padding last


def test_axis_angle_call_hand_pca_and_mean_pose_known_answers():
    """pose2rot=True restatement (smplx 0.1.28 SMPLX.forward): with flat_hand_mean and no hand input the hands are identity,
    i.e. the rotation-matrix call on batch_rodrigues of the body pose; without flat_hand_mean an un-supplied hand IS the
    model file's mean hand pose; PCA coefficients act through the first num_pca_comps rows of hands_components."""
    from airpose_amd import smplx_model as SM
    md = SM.make_synthetic_model(4321, num_verts=600, num_faces=900)
    gen = torch.Generator().manual_seed(1)
    B = 2
    betas, bp = torch.randn(B, 10, generator=gen), torch.randn(B, 63, generator=gen) * 0.4
    rm = smplx_ref.batch_rodrigues(bp.reshape(-1, 3)).reshape(B, 21, 3, 3)
    v_rot, j_rot = smplx_ref.smplx_forward(md, betas, rm)
    v_flat, j_flat = smplx_ref.smplx_forward_axis_angle(md, betas, bp, flat_hand_mean=True)
    assert torch.allclose(v_flat, v_rot, atol=1e-6) and torch.allclose(j_flat, j_rot, atol=1e-6)
    # un-supplied hands, non-flat mean == explicit mean hand pose through the flat axis-angle convention
    v_mean, _ = smplx_ref.smplx_forward_axis_angle(md, betas, bp)
    ml, mr = torch.from_numpy(md["hands_meanl"]).expand(B, 45), torch.from_numpy(md["hands_meanr"]).expand(B, 45)
    v_exp, _ = smplx_ref.smplx_forward_axis_angle(md, betas, bp, left_hand_pose=ml, right_hand_pose=mr, use_pca=False,
                                                  flat_hand_mean=True)
    assert torch.allclose(v_mean, v_exp, atol=1e-6) and (v_mean - v_flat).abs().max() > 1e-3
    # 6 PCA coefficients == their expansion handed over as 45 axis-angle numbers
    c = torch.randn(B, 6, generator=gen)
    v_pca, _ = smplx_ref.smplx_forward_axis_angle(md, betas, bp, left_hand_pose=c, flat_hand_mean=True)
    full = c @ torch.from_numpy(md["hands_componentsl"][:6])
    v_full, _ = smplx_ref.smplx_forward_axis_angle(md, betas, bp, left_hand_pose=full, use_pca=False, flat_hand_mean=True)
    assert torch.allclose(v_pca, v_full, atol=1e-6)
