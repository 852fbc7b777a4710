"""ORACLE (test infrastructure only).  CPU restatement of the SMPL-X forward pass.

PARITY UNPINNED.  The arithmetic lives in the third-party git submodule
``nitin-ppnp/smplx`` (a modified fork of ``vchoutas/smplx``; author's env pins
``smplx==0.1.28``, copenet/requirements.txt:328).  The submodule is EMPTY in
/root/reference (.gitmodules:5-10) and the licence-gated model files are absent, so
neither the code nor golden outputs are available.  This file restates the
*published* upstream algorithm (smplx 0.1.28: ``lbs.lbs``, ``blend_shapes``,
``vertices2joints``, ``batch_rigid_transform``, ``transform_mat``,
``vertices2landmarks``; ``body_models.SMPLX.forward``; ``VertexJointSelector``) and is
anchored on the reference's call sites:
  copenet/src/copenet/copenet_twoview.py:36-45   ctor (batch_size, create_transl=False)
  copenet/src/copenet/copenet_twoview.py:237-241 forward(betas, body_pose (B,21,3,3), global_orient (B,1,3,3),
                                                 transl, pose2rot=False) -> .vertices (B,10475,3) .joints (B,127,3)
  copenet/src/copenet/copenet_twoview.py:589     joints reshaped (-1,4,127,3)
Known-answer tests (tests/test_oracle_smplx.py) are the only pin: identity pose,
shape-only closed form, single-joint rigid rotation, landmark/extra-joint gathers.
Fork delta (inferred from call sites): with pose2rot=False the un-supplied jaw / eye /
hand rotations are identity matrices (upstream's SMPLXLayer behaviour).
"""
import torch
import torch.nn.functional as F


def blend_shapes(betas, shape_disps):
    # upstream lbs.blend_shapes: einsum('bl,mkl->bmk')
    return torch.einsum("bl,mkl->bmk", betas, shape_disps)


def vertices2joints(J_regressor, vertices):
    # upstream lbs.vertices2joints: einsum('bik,ji->bjk')
    return torch.einsum("bik,ji->bjk", vertices, J_regressor)


def batch_rigid_transform(rot_mats, joints, parents):
    """upstream lbs.batch_rigid_transform: root-relative chain, rest pose removed."""
    B, J = joints.shape[:2]
    joints = joints.unsqueeze(-1)
    rel = joints.clone()
    rel[:, 1:] -= joints[:, parents[1:]]
    # transform_mat: [[R, t], [0, 1]]
    tm = torch.cat([F.pad(rot_mats.reshape(-1, 3, 3), [0, 0, 0, 1]),
                    F.pad(rel.reshape(-1, 3, 1), [0, 0, 0, 1], value=1.0)], dim=2).reshape(B, J, 4, 4)
    chain = [tm[:, 0]]
    for i in range(1, J):
        chain.append(torch.matmul(chain[int(parents[i])], tm[:, i]))
    transforms = torch.stack(chain, dim=1)
    posed_joints = transforms[:, :, :3, 3]
    joints_h = F.pad(joints, [0, 0, 0, 1])
    rel_transforms = transforms - F.pad(torch.matmul(transforms, joints_h), [3, 0, 0, 0, 0, 0, 0, 0])
    return posed_joints, rel_transforms


def lbs(betas, rot_mats, v_template, shapedirs, posedirs, J_regressor, parents, lbs_weights):
    """upstream lbs.lbs with pose2rot=False; rot_mats (B,J,3,3)."""
    B = max(betas.shape[0], rot_mats.shape[0])
    dtype = betas.dtype
    v_shaped = v_template + blend_shapes(betas, shapedirs)
    J = vertices2joints(J_regressor, v_shaped)
    ident = torch.eye(3, dtype=dtype)
    pose_feature = (rot_mats[:, 1:] - ident).reshape(B, -1)
    v_posed = torch.matmul(pose_feature, posedirs).view(B, -1, 3) + v_shaped
    J_transformed, A = batch_rigid_transform(rot_mats, J, parents)
    W = lbs_weights.unsqueeze(0).expand(B, -1, -1)
    T = torch.matmul(W, A.view(B, -1, 16)).view(B, -1, 4, 4)
    v_h = torch.cat([v_posed, torch.ones(B, v_posed.shape[1], 1, dtype=dtype)], dim=2)
    verts = torch.matmul(T, v_h.unsqueeze(-1))[:, :, :3, 0]
    return verts, J_transformed


def vertices2landmarks(vertices, faces, lmk_faces_idx, lmk_bary_coords):
    lmk_faces = faces[lmk_faces_idx]                   # (L,3)
    lmk_vertices = vertices[:, lmk_faces]              # (B,L,3,3)
    return torch.einsum("blfi,lf->bli", lmk_vertices, lmk_bary_coords)


def smplx_forward(model, betas, body_pose, global_orient=None, transl=None, expression=None,
                  jaw_pose=None, leye_pose=None, reye_pose=None, left_hand_pose=None,
                  right_hand_pose=None, dtype=torch.float32):
    """SMPLX.forward(..., pose2rot=False) -> (vertices (B,V,3), joints (B,127,3)).

    `model` is the dict of numpy arrays from airpose_amd.smplx_model (data only)."""
    t = lambda a: torch.as_tensor(a).to(dtype)
    B = betas.shape[0]
    eye = torch.eye(3, dtype=dtype).expand(B, 1, 3, 3)
    go = eye if global_orient is None else global_orient.reshape(B, 1, 3, 3).to(dtype)
    body = body_pose.reshape(B, 21, 3, 3).to(dtype)
    jaw = eye if jaw_pose is None else jaw_pose.reshape(B, 1, 3, 3).to(dtype)
    le = eye if leye_pose is None else leye_pose.reshape(B, 1, 3, 3).to(dtype)
    re = eye if reye_pose is None else reye_pose.reshape(B, 1, 3, 3).to(dtype)
    lh = eye.expand(B, 15, 3, 3) if left_hand_pose is None else left_hand_pose.reshape(B, 15, 3, 3).to(dtype)
    rh = eye.expand(B, 15, 3, 3) if right_hand_pose is None else right_hand_pose.reshape(B, 15, 3, 3).to(dtype)
    # upstream order: global, body, jaw, leye, reye, left hand, right hand
    full_pose = torch.cat([go, body, jaw, le, re, lh, rh], dim=1)
    expr = torch.zeros(B, 10, dtype=dtype) if expression is None else expression.to(dtype)
    shape_components = torch.cat([betas.to(dtype), expr], dim=-1)
    parents = torch.as_tensor(model["parents"]).long()
    verts, joints = lbs(shape_components, full_pose, t(model["v_template"]), t(model["shapedirs"]),
                        t(model["posedirs"]), t(model["J_regressor"]), parents, t(model["lbs_weights"]))
    faces = torch.as_tensor(model["faces"]).long()
    landmarks = vertices2landmarks(verts, faces, torch.as_tensor(model["lmk_faces_idx"]).long(),
                                   t(model["lmk_bary_coords"]))
    extra = verts[:, torch.as_tensor(model["extra_joint_verts"]).long()]     # VertexJointSelector
    joints = torch.cat([joints, extra, landmarks], dim=1)
    if transl is not None:
        joints = joints + transl.to(dtype).unsqueeze(1)
        verts = verts + transl.to(dtype).unsqueeze(1)
    return verts, joints


def batch_rodrigues(rot_vecs, epsilon=1e-8):
    """upstream lbs.batch_rodrigues: angle = |r + eps|, K = skew(r / angle), R = I + sin K + (1 - cos) K K."""
    N = rot_vecs.shape[0]
    angle = torch.norm(rot_vecs + epsilon, dim=1, keepdim=True)
    rot_dir = rot_vecs / angle
    cos, sin = torch.cos(angle).unsqueeze(1), torch.sin(angle).unsqueeze(1)
    rx, ry, rz = torch.split(rot_dir, 1, dim=1)
    zeros = torch.zeros(N, 1, dtype=rot_vecs.dtype)
    K = torch.cat([zeros, -rz, ry, rz, zeros, -rx, -ry, rx, zeros], dim=1).view(N, 3, 3)
    ident = torch.eye(3, dtype=rot_vecs.dtype).unsqueeze(0)
    return ident + sin * K + (1 - cos) * torch.bmm(K, K)


def smplx_forward_axis_angle(model, betas, body_pose, global_orient=None, transl=None, expression=None, jaw_pose=None,
                             leye_pose=None, reye_pose=None, left_hand_pose=None, right_hand_pose=None, use_pca=True,
                             num_pca_comps=6, flat_hand_mean=False, dtype=torch.float32):
    """upstream body_models.SMPLX.forward(..., pose2rot=True) of smplx 0.1.28 (the call of the reference's dataset code,
    copenet/src/copenet/dsets/aerialpeople.py:56-64,181-197): un-supplied poses are the module's zero parameters, hands
    are `coeffs @ hands_components[:num_pca_comps]` when use_pca, `full_pose += pose_mean` (mean hand pose of the model file
    unless flat_hand_mean, zero elsewhere), then lbs with batch_rodrigues on every joint."""
    B = betas.shape[0]
    z = lambda n: torch.zeros(B, n, dtype=dtype)
    t = lambda a: torch.as_tensor(a).to(dtype)
    hands = []
    for side, h in (("l", left_hand_pose), ("r", right_hand_pose)):
        h = z(num_pca_comps if use_pca else 45) if h is None else h.to(dtype)
        if use_pca:
            h = torch.einsum("bi,ij->bj", h, t(model["hands_components" + side])[:num_pca_comps])
        hands.append(h + (0 if flat_hand_mean else t(model["hands_mean" + side])))
    full = torch.cat([z(3) if global_orient is None else global_orient.reshape(B, 3).to(dtype), body_pose.reshape(B, 63).to(dtype),
                      z(3) if jaw_pose is None else jaw_pose.reshape(B, 3).to(dtype),
                      z(3) if leye_pose is None else leye_pose.reshape(B, 3).to(dtype),
                      z(3) if reye_pose is None else reye_pose.reshape(B, 3).to(dtype), hands[0], hands[1]], dim=1)
    R = batch_rodrigues(full.reshape(-1, 3)).reshape(B, 55, 3, 3)
    return smplx_forward(model, betas, R[:, 1:22], R[:, :1], transl, expression, R[:, 22:23], R[:, 23:24], R[:, 24:25],
                         R[:, 25:40], R[:, 40:55], dtype=dtype)
