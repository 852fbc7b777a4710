"""ORACLE (test infrastructure, never shipped, never measured as the product).

CPU restatement, in plain torch functional ops on fp32 (or fp64) tensors, of the
reference network on the hot path:

  trunk       copenet.forward_feat_ext   copenet/src/copenet/models/model_copenet.py:161-176
  bottleneck  Bottleneck.forward         model_copenet.py:27-47
  regressor   copenet.forward_reg        model_copenet.py:178-204
  IEF driver  copenet.forward            model_copenet.py:112-159
  hmr head    model_hmr.copenet.forward  copenet/src/copenet/models/model_hmr.py:112-172

Parity pinning: tests/test_oracle_golden.py checks every function here against
tests/golden/*.npz, which tools/make_golden.py produced by importing the real
reference modules from /root/reference in the build container (PINNED).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
Weights come in as a state_dict with the reference's key names.
"""
import torch
import torch.nn.functional as F

LAYERS = (3, 4, 6, 3)
BN_EPS = 1e-5  # nn.BatchNorm2d default, used by model_copenet.py:17,20,22,60


def _bn(x, sd, p):
    # eval-mode BatchNorm2d: (x - mean) / sqrt(var + eps) * gamma + beta
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"],
                        sd[p + ".weight"], sd[p + ".bias"], False, 0.0, BN_EPS)


def bottleneck(x, sd, p, stride, has_down):
    """model_copenet.py:27-47 (stride on the 3x3: line 18)."""
    out = F.relu(_bn(F.conv2d(x, sd[p + ".conv1.weight"]), sd, p + ".bn1"))
    out = F.relu(_bn(F.conv2d(out, sd[p + ".conv2.weight"], stride=stride, padding=1), sd, p + ".bn2"))
    out = _bn(F.conv2d(out, sd[p + ".conv3.weight"]), sd, p + ".bn3")
    if has_down:
        residual = _bn(F.conv2d(x, sd[p + ".downsample.0.weight"], stride=stride), sd, p + ".downsample.1")
    else:
        residual = x
    return F.relu(out + residual)


def stem(x, sd):
    """model_copenet.py:163-166: conv7x7/2 p3 -> BN -> ReLU -> maxpool3/2 p1."""
    x = F.relu(_bn(F.conv2d(x, sd["conv1.weight"], stride=2, padding=3), sd, "bn1"))
    return F.max_pool2d(x, kernel_size=3, stride=2, padding=1)


def forward_feat_ext(x, sd, taps=None):
    """model_copenet.py:161-176.  `taps` (dict) optionally receives per-stage activations."""
    x = stem(x, sd)
    if taps is not None:
        taps["stem"] = x
    for li, nblocks in enumerate(LAYERS, start=1):
        for bi in range(nblocks):
            stride = 2 if (bi == 0 and li > 1) else 1
            x = bottleneck(x, sd, "layer%d.%d" % (li, bi), stride, bi == 0)
        if taps is not None:
            taps["layer%d" % li] = x
    x = F.avg_pool2d(x, 7, stride=1)           # AvgPool2d(7, stride=1), :173
    return x.view(x.size(0), -1)


def _lin(x, sd, p):
    return F.linear(x, sd[p + ".weight"], sd[p + ".bias"])


def forward_reg(sd, xf0, xf1, bb0, bb1, pos0, pos1, orient0, orient1, art0, art1, shape0, shape1):
    """model_copenet.py:178-204.  Concat order from :185 and :192; dropout is identity in eval."""
    xc0 = torch.cat([xf0, bb0, pos0, orient0, art0, shape0, art1, shape1], 1)
    xc0 = _lin(_lin(xc0, sd, "fc1"), sd, "fc2")
    xc1 = torch.cat([xf1, bb1, pos1, orient1, art1, shape1, art0, shape0], 1)
    xc1 = _lin(_lin(xc1, sd, "fc1"), sd, "fc2")
    pshape0 = shape0 + _lin(xc0, sd, "decshape")
    ppose0 = torch.cat([pos0, orient0, art0], 1) + _lin(xc0, sd, "decpose")
    pshape1 = shape1 + _lin(xc1, sd, "decshape")
    ppose1 = torch.cat([pos1, orient1, art1], 1) + _lin(xc1, sd, "decpose")
    return ppose0, pshape0, ppose1, pshape1


def ief(sd, xf0, xf1, bb0, bb1, init_position0, init_position1,
        init_theta0=None, init_theta1=None, init_shape0=None, init_shape1=None, iters=3):
    """IEF loop of model_copenet.py:119-159 starting from trunk features."""
    B = xf0.shape[0]
    ip = sd["init_pose"]

    def _init(theta):
        t = ip if theta is None else theta
        return t[:, :6].expand(B, -1), t[:, 6:22 * 6].expand(B, -1)

    o0, a0 = _init(init_theta0)
    o1, a1 = _init(init_theta1)
    s0 = sd["init_shape"].expand(B, -1) if init_shape0 is None else init_shape0
    s1 = sd["init_shape"].expand(B, -1) if init_shape1 is None else init_shape1
    p0, b0, p1, b1 = forward_reg(sd, xf0, xf1, bb0, bb1, init_position0, init_position1,
                                 o0, o1, a0, a1, s0, s1)
    for _ in range(int(iters) - 1):
        p0, b0, p1, b1 = forward_reg(sd, xf0, xf1, bb0, bb1, p0[:, :3], p1[:, :3],
                                     p0[:, 3:9], p1[:, 3:9], p0[:, 9:], p1[:, 9:], b0, b1)
    return p0, b0, p1, b1


def sep_forward_reg(sd0, sd1, xf0, xf1, bb0, bb1, pos0, pos1, orient0, orient1, art0, art1, shape0, shape1):
    """copenet_real/models/model_copenet_sep.py:184-210 -- two weight sets; view 1's input is built AFTER `pred_shape0`
    has been rebound to view 0's updated shape (:197-198, 202), so it sees [old art_pose0 | NEW shape0]."""
    xc0 = torch.cat([xf0, bb0, pos0, orient0, art0, shape0, art1, shape1], 1)
    xc0 = _lin(_lin(xc0, sd0, "fc1"), sd0, "fc2")
    pshape0 = shape0 + _lin(xc0, sd0, "decshape")
    ppose0 = torch.cat([pos0, orient0, art0], 1) + _lin(xc0, sd0, "decpose")
    xc1 = torch.cat([xf1, bb1, pos1, orient1, art1, shape1, art0, pshape0], 1)
    xc1 = _lin(_lin(xc1, sd1, "fc1"), sd1, "fc2")
    pshape1 = shape1 + _lin(xc1, sd1, "decshape")
    ppose1 = torch.cat([pos1, orient1, art1], 1) + _lin(xc1, sd1, "decpose")
    return ppose0, pshape0, ppose1, pshape1


def sep_ief(sd0, sd1, xf0, xf1, bb0, bb1, init_position0, init_position1, iters=3):
    """IEF loop of model_copenet_sep.py:137-182 from trunk features (default initial state of each sub-model)."""
    B = xf0.shape[0]
    o0, a0 = sd0["init_pose"][:, :6].expand(B, -1), sd0["init_pose"][:, 6:132].expand(B, -1)
    o1, a1 = sd1["init_pose"][:, :6].expand(B, -1), sd1["init_pose"][:, 6:132].expand(B, -1)
    s0, s1 = sd0["init_shape"].expand(B, -1), sd1["init_shape"].expand(B, -1)
    p0, b0, p1, b1 = sep_forward_reg(sd0, sd1, xf0, xf1, bb0, bb1, init_position0, init_position1, o0, o1, a0, a1, s0, s1)
    for _ in range(int(iters) - 1):
        p0, b0, p1, b1 = sep_forward_reg(sd0, sd1, xf0, xf1, bb0, bb1, p0[:, :3], p1[:, :3], p0[:, 3:9], p1[:, 3:9],
                                         p0[:, 9:], p1[:, 9:], b0, b1)
    return p0, b0, p1, b1


def copenet_forward(sd, x0, x1, bb0, bb1, init_position0, init_position1,
                    init_theta0=None, init_theta1=None, init_shape0=None, init_shape1=None, iters=3):
    """model_copenet.py:112-159."""
    xf0 = forward_feat_ext(x0, sd)
    xf1 = forward_feat_ext(x1, sd)
    return ief(sd, xf0, xf1, bb0, bb1, init_position0, init_position1,
               init_theta0, init_theta1, init_shape0, init_shape1, iters)


# ------------------------------------------------------------------ hmr (Config 1, CPU plumbing)
def singleview_forward(sd, x, bb, init_position, init_theta=None, init_shape=None, iters=3):
    """models/model_copenet_singleview.py:108-138 (+ forward_reg :156-168): xc = [xf | bb | pose135 | shape10]."""
    B = x.shape[0]
    theta = sd["init_pose"][:, :132].expand(B, -1) if init_theta is None else init_theta
    pose = torch.cat([init_position, theta], 1)
    shape = sd["init_shape"].expand(B, -1) if init_shape is None else init_shape
    xf = forward_feat_ext(x, sd)
    for _ in range(int(iters)):
        xc = _lin(_lin(torch.cat([xf, bb, pose, shape], 1), sd, "fc1"), sd, "fc2")
        pose, shape = _lin(xc, sd, "decpose") + pose, _lin(xc, sd, "decshape") + shape
    return pose, shape


def muhmr_forward(sd, x0, x1, iters=3):
    """models/model_muhmr.py:112-199 with the default initial state: xc = [xf | cam | orient | art | shape | partner's
    art, shape] (:168,174); both views are evaluated from the OLD state (symmetric swap)."""
    B = x0.shape[0]
    xf = [forward_feat_ext(x0, sd), forward_feat_ext(x1, sd)]
    pose = [sd["init_pose"][:, :132].expand(B, -1)] * 2
    shape = [sd["init_shape"].expand(B, -1)] * 2
    cam = [sd["init_cam"].expand(B, -1)] * 2
    for _ in range(int(iters)):
        xc = [_lin(_lin(torch.cat([xf[v], cam[v], pose[v], shape[v], pose[1 - v][:, 6:], shape[1 - v]], 1), sd, "fc1"),
                   sd, "fc2") for v in (0, 1)]
        pose = [pose[v] + _lin(xc[v], sd, "decpose") for v in (0, 1)]
        shape = [shape[v] + _lin(xc[v], sd, "decshape") for v in (0, 1)]
        cam = [cam[v] + _lin(xc[v], sd, "deccam") for v in (0, 1)]
    return pose[0], shape[0], cam[0], pose[1], shape[1], cam[1]


def hmr_forward_reg(sd, xf, pose, shape, cam):
    """model_hmr.py:160-172."""
    xc = _lin(_lin(torch.cat([xf, pose, shape, cam], 1), sd, "fc1"), sd, "fc2")
    return _lin(xc, sd, "decpose") + pose, _lin(xc, sd, "decshape") + shape, _lin(xc, sd, "deccam") + cam


def hmr_forward(sd, x, iters=3):
    """model_hmr.py:112-141; returns (rotmat (B,22,3,3), betas, cam)."""
    from .geometry_ref import rot6d_to_rotmat
    B = x.shape[0]
    pose = sd["init_pose"][:, :22 * 6].expand(B, -1)
    shape = sd["init_shape"].expand(B, -1)
    cam = sd["init_cam"].expand(B, -1)
    xf = forward_feat_ext(x, sd)
    for _ in range(int(iters)):
        pose, shape, cam = hmr_forward_reg(sd, xf, pose, shape, cam)
    return rot6d_to_rotmat(pose).view(B, 22, 3, 3), shape, cam
