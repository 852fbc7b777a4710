"""Host mirror of the inference branch of ``copenet_twoview.fwd_pass_and_loss``
(copenet/src/copenet/copenet_twoview.py:166-223, 236-279, 307-350): batch dict in, the reference's
test-mode output dict out.  Python only orchestrates: TWO C-ABI calls per forward -- ap_copenet_fwd and
ap_smplx_fwd_twoview (translation un-scale in place on pred_pose, rot6d, SMPL-X for both views, root transform,
projection with the camera centres read from the intrinsics, optionally the test-mode input meshes) -- and no
torch kernel in between."""
import torch

from . import _native as N

TRANS_SCALE = 0.05               # copenet_twoview.py:199
FOCAL_LENGTH = (1475.0, 1475.0)  # copenet/src/copenet/constants.py:7


class TwoViewInference(object):
    def __init__(self, model, smplx, iters=3, focal_length=FOCAL_LENGTH):
        self.model, self.smplx, self.iters, self.focal_length = model, smplx, iters, focal_length
        self._pos = {}

    def init_position(self, B, device):
        """[0,0,10] * 0.05 (copenet_twoview.py:184-185,201-203), cached per (B, device)."""
        key = (B, device)
        if key not in self._pos:
            self._pos[key] = (torch.tensor([0.0, 0.0, 10.0], device=device).expand(B, -1).clone() * TRANS_SCALE)
        return self._pos[key]

    def forward_net(self, im0, im1, bb0, bb1):
        B, dev = im0.shape[0], im0.device
        pos = self.init_position(B, dev)
        return self.model(x0=im0, x1=im1, bb0=bb0, bb1=bb1, init_position0=pos, init_position1=pos, iters=self.iters)

    def __call__(self, batch, want_rotmat=True, want_angles=False, want_input_mesh=False):
        im0, im1 = batch["im0"], batch["im1"]
        B, dev = im0.shape[0], im0.device
        p0, b0, p1, b1 = self.forward_net(im0, im1, batch["bb0"], batch["bb1"])
        # pred_pose0/1 and betas0/1 are the two halves of one (2,B,.) buffer: both views run as 2B bodies
        pose = p0._base if p0._base is not None and p0._base.shape == (2, B, 135) else torch.stack([p0, p1])
        betas = b0._base if b0._base is not None and b0._base.shape == (2, B, 10) else torch.stack([b0, b1])
        in_trans = None
        if want_input_mesh:
            # in_smpltrans after the reference's ``*= trans_scale`` ... ``/= trans_scale`` round trip (:199-203,216-218)
            key = ("in", B, dev)
            if key not in self._pos:
                t = self.init_position(B, dev) / TRANS_SCALE
                self._pos[key] = torch.stack([t, t]).contiguous()
            in_trans = self._pos[key]
        # ``pred_smpltrans /= trans_scale`` happens inside the native call, in place on pose (so the returned
        # pred_pose is un-scaled too, exactly as in the reference where pred_smpltrans is a view of pred_pose)
        o = self.smplx.forward_twoview(pose, betas, batch["intr0"], batch["intr1"], trans_scale=TRANS_SCALE,
                                       in_smpltrans=in_trans, focal_length=self.focal_length, want_rotmat=want_rotmat)
        out = {}
        for v in (0, 1):
            out["pred_pose%d" % v] = pose[v]
            out["pred_betas%d" % v] = betas[v]
            out["pred_smpltrans%d" % v] = pose[v, :, :3]
            out["pred_vertices_cam%d" % v] = o["vertices_cam"][v]
            out["pred_j3d_cam%d" % v] = o["j3d_cam"][v]
            out["pred_j2d_cam%d" % v] = o["j2d_cam"][v]
            if want_rotmat:
                out["pred_rotmat%d" % v] = o["rotmat"][v]
            if want_input_mesh:                            # :258-279, 330-331, 342-343
                out["pred_vertices_cam_in%d" % v] = o["vertices_cam_in"][v]
                out["in_smpltrans%d" % v] = in_trans[v]
        if want_angles:                                    # test-mode output of the caller, :323-324
            if not want_rotmat:
                raise ValueError("want_angles needs want_rotmat")
            from .geometry import rotation_matrix_to_angle_axis
            ang = rotation_matrix_to_angle_axis(o["rotmat"].reshape(-1, 3, 3)).view(2, B, 22, 3)
            out["pred_angles0"], out["pred_angles1"] = ang[0], ang[1]
        return out
