"""Drop-in for the reference's single-view AirPose baseline network (``--model copenet_singleview``).

Mirrors ``copenet.models.model_copenet_singleview`` (copenet/src/copenet/models/model_copenet_singleview.py): the same
trunk, ``fc1`` input 2048 + 3 (bb) + 135 (pose) + 10 (shape) = 2196, ``forward(x, bb, init_position, init_cam,
init_theta, init_shape, iters)`` (:108-138) -> ``(pred_pose (B,135), pred_betas (B,10))``.  Same state_dict keys as the
reference module; compute through ap_singleview_fwd (libairpose_hip.so): trunk + folded IEF kernel, no CPU path.
"""
import torch

from . import _native as N
from .copenet_model import Bottleneck, copenet as _copenet_base


class copenet(_copenet_base):
    variant = 2
    fc1_extra = 3 + (3 + 22 * 6) + 10

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        del self.init_cam                                   # buffers of the reference module (:86-92), in its order
        self.register_buffer("init_position", torch.tensor([[0.0, 0.0, 10.0 / 0.05]], dtype=torch.float32))

    def forward(self, x, bb, init_position, init_cam=None, init_theta=None, init_shape=None, iters=3):
        self._check_eval()
        dev = self._dev(x)
        if x.dim() != 4 or x.shape[1:] != (3, 224, 224):
            raise RuntimeError("forward expects (B, 3, 224, 224) crops")
        B = x.shape[0]
        x, bb, pos = N.f32c(x), N.f32c(bb, dev), N.f32c(init_position, dev)
        if bb.shape != (B, 3) or pos.shape != (B, 3):
            raise RuntimeError("bb and init_position must be (B, 3)")
        th, ths = self._bs(N.f32c(init_theta, dev), B, 132, "init_theta")
        sh, shs = self._bs(N.f32c(init_shape, dev), B, 10, "init_shape")
        pose = torch.empty(B, 135, device=dev, dtype=torch.float32)
        betas = torch.empty(B, 10, device=dev, dtype=torch.float32)
        with self._lock, torch.cuda.device(dev):
            h = self._native(dev)
            N.check(N.lib().ap_singleview_fwd(h, N.dptr(x), N.dptr(bb), N.dptr(pos), N.dptr(th), ths, N.dptr(sh), shs, B,
                                              int(iters), N.dptr(pose), N.dptr(betas), N.stream_ptr(dev)),
                    "ap_singleview_fwd")
        return pose, betas

    def forward_reg(self, xf, bb, pred_pose, pred_shape, iters=1):
        """One regressor evaluation from trunk features (model_copenet_singleview.py:159-170):
        (xf (B,2048), bb (B,3), pose (B,135) = trans3 | 6-D, shape (B,10)) -> the updated (pose, shape)."""
        self._check_eval()
        dev = self._dev(xf)
        B = xf.shape[0]
        xf, bb, p, s = (N.f32c(t, dev) for t in (xf, bb, pred_pose, pred_shape))
        if xf.shape != (B, 2048) or bb.shape != (B, 3) or p.shape != (B, 135) or s.shape != (B, 10):
            raise RuntimeError("forward_reg: xf (B,2048), bb (B,3), pred_pose (B,135), pred_shape (B,10)")
        pose = torch.empty(B, 135, device=dev, dtype=torch.float32)
        betas = torch.empty(B, 10, device=dev, dtype=torch.float32)
        pos, th = p[:, :3].contiguous(), p[:, 3:].contiguous()
        with self._lock, torch.cuda.device(dev):
            h = self._native(dev)
            N.check(N.lib().ap_singleview_reg(h, N.dptr(xf), N.dptr(bb), N.dptr(pos), N.dptr(th), 132, N.dptr(s), 10, B,
                                              int(iters), N.dptr(pose), N.dptr(betas), N.stream_ptr(dev)),
                    "ap_singleview_reg")
        return pose, betas

    def forward_ief(self, *a, **k):
        raise NotImplementedError("model_copenet_singleview has no two-view IEF entry; use forward() or forward_reg()")

    regressor_step = forward_ief


def getcopenet(smpl_mean_params, pretrained=True, precision="f16", **kwargs):
    """model_copenet_singleview.getcopenet; weights arrive through load_state_dict (no torchvision / network here)."""
    return copenet(Bottleneck, [3, 4, 6, 3], smpl_mean_params, precision=precision, **kwargs)
