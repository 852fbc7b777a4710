"""Host mirror of the inference branch of ``copenet_twoview.fwd_pass_and_loss``
(copenet/src/copenet/copenet_twoview.py:166-223, 236-257, 307-350): batch dict in, the reference's
test-mode output dict out.  Python only orchestrates: four C-ABI calls per forward
(ap_copenet_fwd, one torch in-place un-scale of the translation exactly as the reference does it,
ap_smplx_fwd_fused for both views at once)."""
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

    def __call__(self, batch, want_rotmat=True, want_angles=False):
        im0, im1 = batch["im0"], batch["im1"]
        B, dev = im0.shape[0], im0.device
        p0, b0, p1, b1 = self.forward_net(im0, im1, batch["bb0"], batch["bb1"])
        # pred_pose0/1 and betas0/1 are the two halves of one (2,B,.) buffer: run both views as 2B bodies
        pose = p0._base if p0._base is not None and p0._base.shape == (2, B, 135) else torch.stack([p0, p1])
        betas = b0._base if b0._base is not None and b0._base.shape == (2, B, 10) else torch.stack([b0, b1])
        pose[:, :, :3] /= TRANS_SCALE                      # :214-218, in place: pred_pose itself is un-scaled
        cc = torch.cat([N.f32c(batch["intr0"], dev)[:, :2, 2], N.f32c(batch["intr1"], dev)[:, :2, 2]], 0).contiguous()
        o = self.smplx.forward_fused(pose.view(2 * B, 135), betas.view(2 * B, 10), cc, self.focal_length, want_rotmat)
        out = {}
        for v in (0, 1):
            sl = slice(v * B, (v + 1) * B)
            out["pred_pose%d" % v] = pose[v]
            out["pred_betas%d" % v] = betas[v]
            out["pred_smpltrans%d" % v] = pose[v, :, :3]
            out["pred_vertices_cam%d" % v] = o["vertices_cam"][sl]
            out["pred_j3d_cam%d" % v] = o["j3d_cam"][sl]
            out["pred_j2d_cam%d" % v] = o["j2d_cam"][sl]
            if want_rotmat:
                out["pred_rotmat%d" % v] = o["rotmat"][sl]
        if want_angles:                                    # test-mode output of the caller, :323-324
            if not want_rotmat:
                raise ValueError("want_angles needs want_rotmat")
            from .geometry import rotation_matrix_to_angle_axis
            ang = rotation_matrix_to_angle_axis(o["rotmat"].reshape(-1, 3, 3)).view(2, B, 22, 3)
            out["pred_angles0"], out["pred_angles1"] = ang[0], ang[1]
        return out
