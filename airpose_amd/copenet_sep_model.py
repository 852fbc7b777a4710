"""Drop-in for copenet_real/src/copenet_real/models/model_copenet_sep.py: the two-weight-set variant used on the real
(two-drone) data -- one `copenet` per view, cross-view fusion in `forward_reg`.

Reference semantics kept to the letter (model_copenet_sep.py:133-216):
  * every view has its own trunk AND its own regressor head (`copenet0`, `copenet1`; state_dict keys `copenet{0,1}.*`);
  * `forward_reg` evaluates view 0 first and view 1 then sees view 0's ALREADY UPDATED shape next to its OLD
    articulated pose (:197-205: `pred_shape0` is rebound before `xc1` is concatenated) -- a Gauss-Seidel step on the
    shape, not the symmetric swap of the shared-weight model.
Each regressor evaluation is one `ap_regressor_step` (the view-split step of the C ABI: partner state supplied by the
caller), so the same two calls serve the on-drone exchange (README.md:238-241)."""
import torch
import torch.nn as nn

from .copenet_model import Bottleneck, copenet


class copenet_sep(nn.Module):
    def __init__(self, block, layers, smpl_mean_params, precision="f16"):
        super().__init__()
        self.copenet0 = copenet(block, layers, smpl_mean_params, precision=precision)
        self.copenet1 = copenet(block, layers, smpl_mean_params, precision=precision)

    def forward(self, x0, x1, bb0, bb1, init_position0, init_position1, init_theta0=None, init_theta1=None,
                init_shape0=None, init_shape1=None, iters=3):
        xf0 = self.copenet0.forward_feat_ext(x0)            # :163-164
        xf1 = self.copenet1.forward_feat_ext(x1)
        return self.forward_ief(xf0, xf1, bb0, bb1, init_position0, init_position1, init_theta0, init_theta1,
                                init_shape0, init_shape1, iters)

    def forward_ief(self, xf0, xf1, bb0, bb1, init_position0, init_position1, init_theta0=None, init_theta1=None,
                    init_shape0=None, init_shape1=None, iters=3):
        """The IEF loop of forward() from pre-computed trunk features (:144-182)."""
        B = xf0.shape[0]
        dev = xf0.device
        f = lambda t: t.to(device=dev, dtype=torch.float32)

        def init(net, theta, shape):                        # :144-159
            th = net.init_pose if theta is None else theta
            sh = net.init_shape if shape is None else shape
            return f(th)[:, :6].expand(B, -1), f(th)[:, 6:132].expand(B, -1), f(sh).expand(B, -1)
        o0, a0, s0 = init(self.copenet0, init_theta0, init_shape0)
        o1, a1, s1 = init(self.copenet1, init_theta1, init_shape1)
        p0, b0, p1, b1 = self.forward_reg(xf0, xf1, bb0, bb1, f(init_position0), f(init_position1), o0, o1, a0, a1, s0, s1)
        for _ in range(int(iters) - 1):                     # :174-180
            p0, b0, p1, b1 = self.forward_reg(xf0, xf1, bb0, bb1, p0[:, :3], p1[:, :3], p0[:, 3:9], p1[:, 3:9],
                                              p0[:, 9:], p1[:, 9:], b0, b1)
        return p0, b0, p1, b1

    def forward_reg(self, xf0, xf1, bb0, bb1, pred_position0, pred_position1, pred_orient0, pred_orient1,
                    pred_art_pose0, pred_art_pose1, pred_shape0, pred_shape1):
        """model_copenet_sep.py:184-210."""
        pose0 = torch.cat([pred_position0, pred_orient0, pred_art_pose0], 1)
        pose1 = torch.cat([pred_position1, pred_orient1, pred_art_pose1], 1)
        new_pose0, new_shape0 = self.copenet0.regressor_step(xf0, bb0, pose0, pred_shape0,
                                                             torch.cat([pred_art_pose1, pred_shape1], 1))
        # view 1 is fed view 0's OLD articulated pose and its NEW shape (:197-198, 202)
        new_pose1, new_shape1 = self.copenet1.regressor_step(xf1, bb1, pose1, pred_shape1,
                                                             torch.cat([pred_art_pose0, new_shape0], 1))
        return new_pose0, new_shape0, new_pose1, new_shape1


def getcopenet_sep(smpl_mean_params, precision="f16", **kwargs):
    return copenet_sep(Bottleneck, [3, 4, 6, 3], smpl_mean_params, precision=precision, **kwargs)
