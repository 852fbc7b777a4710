"""ORACLE (test infrastructure only).  CPU restatement of the inference branch of
``copenet_twoview.fwd_pass_and_loss`` (copenet/src/copenet/copenet_twoview.py:166-223, 236-257,
307-350): init translation -> network -> un-scale translation (in place on a view of pred_pose)
-> rot6d -> SMPL-X -> transform_smpl -> perspective projection.

Pinned through its parts: network + geometry against tests/golden (PINNED); SMPL-X leg
against known-answer tests only (PARITY UNPINNED, see smplx_ref.py).
"""
import torch

from . import copenet_ref, geometry_ref, smplx_ref

TRANS_SCALE = 0.05               # copenet_twoview.py:199
FOCAL_LENGTH = (1475.0, 1475.0)  # copenet/src/copenet/constants.py:7


def init_position(batch):
    """copenet_twoview.py:184-185,201-203: [0,0,10] * 0.05."""
    return torch.tensor([0.0, 0.0, 10.0]).expand(batch, -1).clone() * TRANS_SCALE


def body_outputs(model, pred_pose, pred_betas, intr, focal=FOCAL_LENGTH, dtype=torch.float32):
    """Rows 5-8 for one view.  pred_pose already has its translation un-scaled."""
    B = pred_pose.shape[0]
    trans = pred_pose[:, :3]
    rotmat = geometry_ref.rot6d_to_rotmat(pred_pose[:, 3:]).view(B, 22, 3, 3)        # :222
    verts, joints = smplx_ref.smplx_forward(model, pred_betas, rotmat[:, 1:],          # :237-241
                                            global_orient=torch.eye(3).expand(B, 1, 3, 3),
                                            transl=torch.zeros(B, 3), dtype=dtype)
    tm = torch.cat([rotmat[:, 0].to(dtype), trans.to(dtype).unsqueeze(2)], dim=2)    # :242-243
    v_cam, j_cam = geometry_ref.transform_smpl(tm, verts, joints)                      # :244-246
    j2d = geometry_ref.perspective_projection(                                         # :307-311
        j_cam, torch.eye(3, dtype=dtype).expand(B, 3, 3), torch.zeros(B, 3, dtype=dtype), focal,
        intr[:, :2, 2].to(dtype).unsqueeze(0))
    return dict(rotmat=rotmat, vertices_cam=v_cam, j3d_cam=j_cam, j2d_cam=j2d, smpltrans=trans)


def input_mesh(model, rotmat, in_smpltrans, dtype=torch.float32):
    """copenet_twoview.py:258-279 (test mode): betas = 0, the predicted body rotations, identity global orientation,
    then transform_smpl([I | in_smpltrans]) -> pred_vertices_cam_in."""
    B = rotmat.shape[0]
    verts, joints = smplx_ref.smplx_forward(model, torch.zeros(B, 10, dtype=dtype), rotmat[:, 1:],
                                            global_orient=torch.eye(3).expand(B, 1, 3, 3),
                                            transl=torch.zeros(B, 3), dtype=dtype)
    tm = torch.cat([torch.eye(3, dtype=dtype).expand(B, 3, 3), in_smpltrans.to(dtype).unsqueeze(2)], dim=2)
    v_cam, _ = geometry_ref.transform_smpl(tm, verts, joints)
    return v_cam


def infer(sd, model, im0, im1, bb0, bb1, intr0, intr1, iters=3, want_input_mesh=False):
    """Whole hot path on CPU; returns the reference's test-mode output dict (tensor subset)."""
    B = im0.shape[0]
    p0, b0, p1, b1 = copenet_ref.copenet_forward(sd, im0.float(), im1.float(), bb0, bb1,
                                                 init_position(B), init_position(B), iters=iters)
    p0, p1 = p0.clone(), p1.clone()
    p0[:, :3] /= TRANS_SCALE          # :214-218, in place on a view: pred_pose itself changes
    p1[:, :3] /= TRANS_SCALE
    out = {"pred_pose0": p0, "pred_pose1": p1, "pred_betas0": b0, "pred_betas1": b1}
    for v, (p, b, intr) in enumerate(((p0, b0, intr0), (p1, b1, intr1))):
        o = body_outputs(model, p, b, intr)
        out["pred_vertices_cam%d" % v] = o["vertices_cam"]
        out["pred_j3d_cam%d" % v] = o["j3d_cam"]
        out["pred_j2d_cam%d" % v] = o["j2d_cam"]
        out["pred_smpltrans%d" % v] = o["smpltrans"]
        out["pred_rotmat%d" % v] = o["rotmat"]
        # :323-324 (test mode): tgm.rotation_matrix_to_angle_axis of the zero-padded (22B,3,4) rotation matrices
        rm = o["rotmat"].reshape(-1, 3, 3)
        out["pred_angles%d" % v] = geometry_ref.rotation_matrix_to_angle_axis(
            torch.cat([rm, torch.zeros(rm.shape[0], 3, 1, dtype=rm.dtype)], 2)).view(B, 22, 3)
        if want_input_mesh:
            # in_smpltrans = [0,0,10] * trans_scale (:184-203) ... /= trans_scale (:216-218) before it is used at :264
            in_t = init_position(B) / TRANS_SCALE
            out["in_smpltrans%d" % v] = in_t
            out["pred_vertices_cam_in%d" % v] = input_mesh(model, o["rotmat"], in_t)
    return out
