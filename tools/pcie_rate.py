#!/usr/bin/env python
"""PCIe-inclusive rate of the hot path (DESIGN note; never the bench `value`): the reference hands the network HOST
tensors (DataLoader output).  Measures, per step of B = 256 pairs:
  (a) fp32 normalised crops (2 x B x 3 x 224 x 224 floats, pinned) -> device, then the forward;
  (b) uint8 224x224 RGB crops -> device, normalised on the GPU by ap_preprocess_crops, then the forward.
Copies are issued on the compute stream (no overlap with the previous step): the conservative, sequential figure."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from airpose_amd import copenet_model, weights as W  # noqa: E402
from airpose_amd.utils import preprocess_crops  # noqa: E402

B = 256
dev = torch.device("cuda", 0)
mp = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "airpose_amd", "data", "smpl_mean_params.npz")
net = copenet_model.getcopenet(mp, precision="bf16")
net.load_state_dict(W.to_torch(W.copenet_state_dict(20240901, mp)))
net.eval().to(dev)
inp = {k: torch.from_numpy(v) for k, v in W.synthetic_inputs(1234, B).items()}
bb0, bb1 = inp["bb0"].to(dev), inp["bb1"].to(dev)
pos = (torch.tensor([0.0, 0.0, 10.0]) * 0.05).expand(B, -1).contiguous().to(dev)
h32 = [inp["im0"].pin_memory(), inp["im1"].pin_memory()]
hu8 = [(torch.rand(B, 224, 224, 3) * 255).to(torch.uint8).pin_memory() for _ in range(2)]
crops = torch.tensor([[0, 224, 0, 224]] * B, dtype=torch.int32, device=dev)


def fwd(x0, x1):
    return net(x0=x0, x1=x1, bb0=bb0, bb1=bb1, init_position0=pos, init_position1=pos, iters=3)


def resident():
    return fwd(d0, d1)


def from_fp32_host():
    return fwd(h32[0].to(dev, non_blocking=True), h32[1].to(dev, non_blocking=True))


def from_uint8_host():
    x = [preprocess_crops(h.to(dev, non_blocking=True), crops, bgr=False)[0] for h in hu8]
    return fwd(x[0], x[1])


d0, d1 = h32[0].to(dev), h32[1].to(dev)
for name, fn in (("inputs resident in HBM", resident), ("fp32 crops from pinned host memory", from_fp32_host),
                 ("uint8 crops from host + GPU normalisation", from_uint8_host)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 100
    print("%-44s %7.2f ms/step  %8.0f pairs/s (network only, B = %d)" % (name, ms, B / ms * 1e3, B))
