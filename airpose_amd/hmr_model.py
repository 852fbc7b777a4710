"""Drop-in for the reference's single-view HMR baseline network (BASELINE config 0's model).

Mirrors ``copenet.models.model_hmr`` (copenet/src/copenet/models/model_hmr.py): same trunk, fc1 input
2048 + 132 + 10 + 3 = 2193, decpose -> 132, weak-perspective ``deccam``; ``forward(x, init_cam, init_theta,
init_shape, iters)`` (:112-141) returns ``(rotmat (B,22,3,3), betas (B,10), cam (B,3))``.  Same state_dict keys as
the reference module; compute through ap_hmr_fwd (libairpose_hip.so), no CPU path.
"""
import torch
import torch.nn as nn

from . import _native as N
from .copenet_model import Bottleneck, copenet as _copenet_base


class copenet(_copenet_base):
    variant = 1
    fc1_extra = 22 * 6 + 10 + 3

    @staticmethod
    def _npose_out(npose):
        return 22 * 6

    def forward(self, x, init_cam=None, init_theta=None, init_shape=None, iters=3):
        self._check_eval()
        dev = self._dev(x)
        if x.dim() != 4 or x.shape[1:] != (3, 224, 224):
            raise RuntimeError("forward expects (B, 3, 224, 224) crops")
        B = x.shape[0]
        x = N.f32c(x)
        th, ths = self._bs(N.f32c(init_theta, dev), B, 132, "init_theta")
        sh, shs = self._bs(N.f32c(init_shape, dev), B, 10, "init_shape")
        cm, cms = self._bs(N.f32c(init_cam, dev), B, 3, "init_cam")
        rot = torch.empty(B, 22, 3, 3, device=dev, dtype=torch.float32)
        betas = torch.empty(B, 10, device=dev, dtype=torch.float32)
        cam = torch.empty(B, 3, device=dev, dtype=torch.float32)
        with self._lock, torch.cuda.device(dev):
            h = self._native(dev)
            N.check(self._L().ap_hmr_fwd(h, N.dptr(x), B, int(iters), N.dptr(th), ths, N.dptr(sh), shs, N.dptr(cm), cms,
                                       N.dptr(rot), N.dptr(betas), N.dptr(cam), N.stream_ptr(dev)), "ap_hmr_fwd")
        return rot, betas, cam

    def forward_reg(self, xf, pred_pose, pred_shape, pred_cam, iters=1):
        """One regressor evaluation from trunk features (model_hmr.py:160-172): (xf (B,2048), pose (B,132) 6-D,
        shape (B,10), cam (B,3)) -> the updated (pose, shape, cam)."""
        self._check_eval()
        dev = self._dev(xf)
        B = xf.shape[0]
        xf, p, s, c = (N.f32c(t, dev) for t in (xf, pred_pose, pred_shape, pred_cam))
        if xf.shape != (B, 2048) or p.shape != (B, 132) or s.shape != (B, 10) or c.shape != (B, 3):
            raise RuntimeError("forward_reg: xf (B,2048), pred_pose (B,132), pred_shape (B,10), pred_cam (B,3)")
        po, so, co = (torch.empty(B, n, device=dev, dtype=torch.float32) for n in (132, 10, 3))
        with self._lock, torch.cuda.device(dev):
            h = self._native(dev)
            N.check(self._L().ap_hmr_reg(h, N.dptr(xf), B, int(iters), N.dptr(p), 132, N.dptr(s), 10, N.dptr(c), 3, N.dptr(po),
                                       N.dptr(so), N.dptr(co), N.stream_ptr(dev)), "ap_hmr_reg")
        return po, so, co

    def forward_ief(self, *a, **k):
        raise NotImplementedError("model_hmr has no two-view IEF entry; use forward() or forward_reg()")

    regressor_step = forward_ief


def getcopenet(smpl_mean_params, pretrained=True, precision="f16", **kwargs):
    """model_hmr.getcopenet; weights arrive through load_state_dict (no torchvision / network here)."""
    return copenet(Bottleneck, [3, 4, 6, 3], smpl_mean_params, precision=precision, **kwargs)
