"""Drop-in for the reference's multi-view HMR baseline network (``--model muhmr``).

Mirrors ``copenet.models.model_muhmr`` (copenet/src/copenet/models/model_muhmr.py): the two-view model with a
weak-perspective camera instead of bounding box + translation -- ``fc1`` input 2048 + 3 + 132 + 10 + 136 = 2329,
``decpose`` -> 132, ``deccam``; ``forward(x0, x1, init_cam0, init_cam1, init_theta0, init_theta1, init_shape0,
init_shape1, iters)`` (:112-161) -> ``(pred_pose0 (B,132), pred_betas0, pred_cam0, pred_pose1, pred_betas1, pred_cam1)``.
Same state_dict keys as the reference module; compute through ap_muhmr_fwd (libairpose_hip.so), no CPU path.
"""
import torch

from . import _native as N
from .copenet_model import Bottleneck, copenet as _copenet_base


class copenet(_copenet_base):
    variant = 3
    fc1_extra = 3 + 22 * 6 + 10 + 21 * 6 + 10

    @staticmethod
    def _npose_out(npose):
        return 22 * 6

    def forward(self, x0, x1, init_cam0=None, init_cam1=None, init_theta0=None, init_theta1=None, init_shape0=None,
                init_shape1=None, iters=3):
        self._check_eval()
        dev = self._dev(x0)
        if x0.dim() != 4 or x0.shape[1:] != (3, 224, 224) or x1.shape != x0.shape:
            raise RuntimeError("forward expects two (B, 3, 224, 224) crops")
        B = x0.shape[0]
        x0, x1 = N.f32c(x0), N.f32c(x1, dev)
        if (init_cam0 is None) != (init_cam1 is None):
            raise RuntimeError("give both initial cameras or neither")
        c0, c0s = self._bs(N.f32c(init_cam0, dev), B, 3, "init_cam0")
        c1, c1s = self._bs(N.f32c(init_cam1, dev), B, 3, "init_cam1")
        if init_cam0 is not None and c0s != c1s:           # one stride for both views in the C ABI
            c0, c1 = c0.expand(B, 3).contiguous(), c1.expand(B, 3).contiguous()
            c0s = c1s = 3
        t0, t0s = self._bs(N.f32c(init_theta0, dev), B, 132, "init_theta0")
        t1, t1s = self._bs(N.f32c(init_theta1, dev), B, 132, "init_theta1")
        s0, s0s = self._bs(N.f32c(init_shape0, dev), B, 10, "init_shape0")
        s1, s1s = self._bs(N.f32c(init_shape1, dev), B, 10, "init_shape1")
        out = torch.empty(2, B, 135, device=dev, dtype=torch.float32)
        betas = torch.empty(2, B, 10, device=dev, dtype=torch.float32)
        with self._lock, torch.cuda.device(dev):
            h = self._native(dev)
            N.check(N.lib().ap_muhmr_fwd(h, N.dptr(x0), N.dptr(x1), N.dptr(c0), c0s, N.dptr(c1), c1s, N.dptr(t0), t0s,
                                         N.dptr(t1), t1s, N.dptr(s0), s0s, N.dptr(s1), s1s, B, int(iters), N.dptr(out[0]),
                                         N.dptr(betas[0]), N.dptr(out[1]), N.dptr(betas[1]), N.stream_ptr(dev)),
                    "ap_muhmr_fwd")
        return out[0, :, 3:], betas[0], out[0, :, :3], out[1, :, 3:], betas[1], out[1, :, :3]

    def forward_reg(self, xf0, xf1, pred_orient0, pred_orient1, pred_art_pose0, pred_art_pose1, pred_shape0, pred_shape1,
                    pred_cam0, pred_cam1):
        """One regressor evaluation for both views from trunk features (model_muhmr.py:177-203) ->
        (pred_pose0 (B,132), pred_shape0, pred_cam0, pred_pose1, pred_shape1, pred_cam1).  Runs the two-view kernels
        with the cameras in the translation slots (the re-mapped fc1 of this variant gives bb zero weight)."""
        self._check_eval()
        dev = self._dev(xf0)
        B = xf0.shape[0]
        xf0, xf1 = N.f32c(xf0), N.f32c(xf1, dev)
        th0 = torch.cat([N.f32c(pred_orient0, dev), N.f32c(pred_art_pose0, dev)], 1).contiguous()
        th1 = torch.cat([N.f32c(pred_orient1, dev), N.f32c(pred_art_pose1, dev)], 1).contiguous()
        s0, s1, c0, c1 = (N.f32c(t, dev) for t in (pred_shape0, pred_shape1, pred_cam0, pred_cam1))
        if xf0.shape != (B, 2048) or xf1.shape != (B, 2048) or th0.shape != (B, 132) or th1.shape != (B, 132) \
                or s0.shape != (B, 10) or s1.shape != (B, 10) or c0.shape != (B, 3) or c1.shape != (B, 3):
            raise RuntimeError("forward_reg: xf (B,2048), orient (B,6), art_pose (B,126), shape (B,10), cam (B,3) per view")
        out = torch.empty(2, B, 135, device=dev, dtype=torch.float32)
        betas = torch.empty(2, B, 10, device=dev, dtype=torch.float32)
        with self._lock, torch.cuda.device(dev):
            h = self._native(dev)
            N.check(N.lib().ap_regressor_fwd(h, N.dptr(xf0), N.dptr(xf1), N.dptr(xf0), N.dptr(xf1), N.dptr(c0), N.dptr(c1),
                                             N.dptr(th0), 132, N.dptr(th1), 132, N.dptr(s0), 10, N.dptr(s1), 10, B, 1,
                                             N.dptr(out[0]), N.dptr(betas[0]), N.dptr(out[1]), N.dptr(betas[1]),
                                             N.stream_ptr(dev)), "ap_regressor_fwd")
        return out[0, :, 3:], betas[0], out[0, :, :3], out[1, :, 3:], betas[1], out[1, :, :3]

    def forward_ief(self, *a, **k):
        raise NotImplementedError("model_muhmr has no translation-based IEF entry; use forward() or forward_reg()")

    regressor_step = forward_ief


def getcopenet(smpl_mean_params, pretrained=True, precision="f16", **kwargs):
    """model_muhmr.getcopenet; weights arrive through load_state_dict (no torchvision / network here)."""
    return copenet(Bottleneck, [3, 4, 6, 3], smpl_mean_params, precision=precision, **kwargs)
