#!/usr/bin/env python
"""Soak of the 16-bit trunk with every hand-counted kernel on (bottleneck2, pair, slab, lean, fused stem, image-resident layer3 blocks,
half-image-resident layer2 3x3, conv_pw): N forward passes over
512 images (256 per view, two concurrent passes), each compared bit for bit with the separate-convolution path of the first
pass.  A rare miss of a counted wait shows up as a mismatch.   python tools/probes/soak_trunk.py [iterations] [f16|bf16]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from airpose_amd import copenet_model, weights as W  # noqa: E402

it = int(sys.argv[1]) if len(sys.argv) > 1 else 200
prec = sys.argv[2] if len(sys.argv) > 2 else "f16"          # "f16" / "bf16": the storage type of the throughput kernels
dev = torch.device("cuda", 0)
mean = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "airpose_amd", "data", "smpl_mean_params.npz")
net = copenet_model.getcopenet(mean, precision=prec).eval()
net.load_state_dict(W.to_torch(W.copenet_state_dict(3, mean)))
g = torch.Generator().manual_seed(1)
x = torch.randn(512, 3, 224, 224, generator=g).to(dev)
def reference(on):
    """on = False: the separate-convolution path (no fused block, pair, image-resident or one-wave-per-SIMD kernel)."""
    net.set_fuse_block(int(on)); net.set_fuse_pair(int(on)); net.set_img_block(int(on)); net.set_img3(int(on)); net.set_pw_conv(int(on))


reference(False)
ref = net.forward_feat_ext(x).clone()
reference(True)
bad = 0
for i in range(it):
    y = net.forward_feat_ext(x)
    if not torch.equal(y, ref):
        bad += 1
        print("iteration %d: %d of %d features differ" % (i, (y != ref).sum().item(), y.numel()))
print("soak, one pass of 512 images: %d iterations, %d mismatching" % (it, bad))
# the two-view forward: 256 pairs = two concurrent passes on two internal streams (+ regressor)
B = 256
bb = torch.rand(B, 3, generator=g).to(dev)
pos = torch.zeros(B, 3, device=dev)
reference(False)
ref2 = [t.clone() for t in net(x[:B], x[B:], bb, bb, pos, pos, iters=3)]
reference(True)
bad2 = 0
for i in range(it):
    out = net(x[:B], x[B:], bb, bb, pos, pos, iters=3)
    if not all(torch.equal(a, b) for a, b in zip(out, ref2)):
        bad2 += 1
        print("two-view iteration %d differs" % i)
print("soak, two-view forward of 256 pairs: %d iterations, %d mismatching" % (it, bad2))
sys.exit(1 if bad or bad2 else 0)
