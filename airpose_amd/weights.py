"""Seeded synthetic weights for the copenet two-view network.

There is no network access for checkpoints, so tests and bench use random-init
weights of the reference architecture.  The generator is pure numpy
(`RandomState` streams are bit-stable across machines) and yields a
``state_dict`` whose keys/shapes are exactly those of the reference module
(reference: copenet/src/copenet/models/model_copenet.py:53-92, torchvision
ResNet-50 naming; 331 entries incl. BN ``num_batches_tracked``).

Distributions follow the reference init (model_copenet.py:74-84) except that
BN statistics are randomised so that the BN scale/shift epilogue is actually
exercised (the reference defaults 0/1/1/0 would hide folding bugs):
  conv      N(0, sqrt(2 / (k*k*C_out)))
  bn        gamma~U(.5,1.5) (last BN of a block: U(.25,.75)), beta~N(0,.1),
            running_mean~N(0,.1), running_var~U(.5,1.5)
            bn="survey": gamma~U(.5,1.5) on EVERY BatchNorm, exactly SURVEY 8(d)'s recipe (the residual
            branches then add at full gain through all 16 blocks); bn="wide": gamma, var~U(.25,2)
  fc1, fc2  U(-1/sqrt(in), 1/sqrt(in)) weight and bias (nn.Linear default)
  dec*      xavier_uniform(gain=0.01) weight, Linear-default bias
"""
import math
from collections import OrderedDict

import numpy as np

LAYERS = (3, 4, 6, 3)            # model_copenet.py:234
PLANES = (64, 128, 256, 512)
EXPANSION = 4                    # model_copenet.py:12
NPOSE = 21 * 6                   # model_copenet.py:56
FC1_IN = 512 * EXPANSION + 3 + 3 + 6 + NPOSE + 10 + NPOSE + 10   # 2332, :67
HMR_FC1_IN = 512 * EXPANSION + 22 * 6 + 10 + 3                  # 2193, model_hmr.py


def _conv(rs, cout, cin, k):
    std = math.sqrt(2.0 / (k * k * cout))
    return rs.normal(0.0, std, size=(cout, cin, k, k)).astype(np.float32)


def _bn(rs, sd, prefix, c, last=False, wide=False):
    # wide: the second statistics range of the parity sweeps (gamma, var ~ U(.25, 2): scales from 0.18 to 4 per channel instead of
    # 0.4 to 2.1; the last BN of a block keeps half the gamma range so that 16 residual blocks do not blow the activations up)
    # wide == "survey": SURVEY 8(d) to the letter -- gamma ~ U(.5, 1.5) on every BatchNorm, the last one of a block included
    if wide == "survey":
        last, wide = False, False
    lo, hi = ((0.125, 1.0) if last else (0.25, 2.0)) if wide else ((0.25, 0.75) if last else (0.5, 1.5))
    sd[prefix + ".weight"] = rs.uniform(lo, hi, size=c).astype(np.float32)
    sd[prefix + ".bias"] = rs.normal(0.0, 0.1, size=c).astype(np.float32)
    sd[prefix + ".running_mean"] = rs.normal(0.0, 0.1, size=c).astype(np.float32)
    vlo, vhi = (0.25, 2.0) if wide else (0.5, 1.5)
    sd[prefix + ".running_var"] = rs.uniform(vlo, vhi, size=c).astype(np.float32)
    sd[prefix + ".num_batches_tracked"] = np.zeros((), dtype=np.int64)


def _linear(rs, sd, prefix, cout, cin, xavier_gain=None):
    if xavier_gain is None:
        b = 1.0 / math.sqrt(cin)
    else:
        b = xavier_gain * math.sqrt(6.0 / (cin + cout))
    sd[prefix + ".weight"] = rs.uniform(-b, b, size=(cout, cin)).astype(np.float32)
    bb = 1.0 / math.sqrt(cin)
    sd[prefix + ".bias"] = rs.uniform(-bb, bb, size=cout).astype(np.float32)


def trunk_state_dict(rs, wide_bn=False):
    """ResNet-50 v1.5 trunk entries (conv1/bn1/layer1..4), reference order."""
    sd = OrderedDict()
    sd["conv1.weight"] = _conv(rs, 64, 3, 7)
    _bn(rs, sd, "bn1", 64, wide=wide_bn)
    inplanes = 64
    for li, (planes, nblocks) in enumerate(zip(PLANES, LAYERS), start=1):
        for bi in range(nblocks):
            p = "layer%d.%d" % (li, bi)
            sd[p + ".conv1.weight"] = _conv(rs, planes, inplanes, 1)
            _bn(rs, sd, p + ".bn1", planes, wide=wide_bn)
            sd[p + ".conv2.weight"] = _conv(rs, planes, planes, 3)
            _bn(rs, sd, p + ".bn2", planes, wide=wide_bn)
            sd[p + ".conv3.weight"] = _conv(rs, planes * EXPANSION, planes, 1)
            _bn(rs, sd, p + ".bn3", planes * EXPANSION, last=True, wide=wide_bn)
            if bi == 0:
                sd[p + ".downsample.0.weight"] = _conv(rs, planes * EXPANSION, inplanes, 1)
                _bn(rs, sd, p + ".downsample.1", planes * EXPANSION, last=True, wide=wide_bn)
            inplanes = planes * EXPANSION
    return sd


def load_mean_params(path):
    d = np.load(path)
    return (d["pose"].astype(np.float32)[None], d["shape"].astype(np.float32)[None],
            d["cam"].astype(np.float32)[None])


def copenet_state_dict(seed, mean_params_path, variant="copenet", wide_bn=False, bn=None):
    """Full state_dict (numpy arrays) for the two-view ``copenet`` (or ``hmr`` / ``copenet_singleview``) module.
    wide_bn: BatchNorm gamma / running_var from U(.25, 2) instead of U(.5, 1.5) (same random stream positions).
    bn: "default" | "wide" (= wide_bn) | "survey" (gamma ~ U(.5, 1.5) on every BatchNorm: SURVEY 8(d) exactly)."""
    rs = np.random.RandomState(seed)
    if bn not in (None, "default", "wide", "survey"):
        raise ValueError(bn)
    sd = trunk_state_dict(rs, "survey" if bn == "survey" else (wide_bn or bn == "wide"))
    if variant not in ("copenet", "hmr", "singleview", "muhmr"):
        raise ValueError(variant)
    fc1_in = {"copenet": FC1_IN, "hmr": HMR_FC1_IN, "singleview": 2048 + 3 + 135 + 10, "muhmr": 2048 + 3 + 132 + 10 + 136}[variant]
    npose_out = 22 * 6 if variant in ("hmr", "muhmr") else 3 + 6 + NPOSE
    _linear(rs, sd, "fc1", 1024, fc1_in)
    _linear(rs, sd, "fc2", 1024, 1024)
    _linear(rs, sd, "decpose", npose_out, 1024, xavier_gain=0.01)
    _linear(rs, sd, "decshape", 10, 1024, xavier_gain=0.01)
    _linear(rs, sd, "deccam", 3, 1024, xavier_gain=0.01)
    pose, shape, cam = load_mean_params(mean_params_path)
    if variant == "singleview":     # model_copenet_singleview.py:89-92 registers init_position instead of init_cam
        sd["init_pose"], sd["init_shape"] = pose, shape
        sd["init_position"] = np.array([[0.0, 0.0, 10.0 / 0.05]], np.float32)
    else:
        sd["init_pose"], sd["init_shape"], sd["init_cam"] = pose, shape, cam
    return sd


def to_torch(sd):
    import torch
    return OrderedDict((k, torch.from_numpy(np.ascontiguousarray(v))) for k, v in sd.items())


def synthetic_inputs(seed, batch, img=224):
    """Synthetic two-view batch: normalised crops, bb, init position, intrinsics."""
    rs = np.random.RandomState(seed)
    out = {}
    for v in (0, 1):
        out["im%d" % v] = rs.standard_normal((batch, 3, img, img)).astype(np.float32)
    for v in (0, 1):
        bb = np.empty((batch, 3), np.float32)
        bb[:, 0:2] = rs.uniform(-0.5, 0.5, size=(batch, 2))
        bb[:, 2] = rs.uniform(0.2, 1.0, size=batch)
        out["bb%d" % v] = bb
    intr = np.zeros((batch, 3, 3), np.float32)
    intr[:, 0, 0] = intr[:, 1, 1] = 1475.0
    intr[:, 0, 2], intr[:, 1, 2], intr[:, 2, 2] = 960.0, 540.0, 1.0
    out["intr0"], out["intr1"] = intr, intr.copy()
    return out
