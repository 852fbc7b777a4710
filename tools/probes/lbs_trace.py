#!/usr/bin/env python
"""Cycle stamps of the fused contraction + skinning kernel (smplx.hip built with -DAP_TRACE): wave 0 of workgroups 0 and 100, their
second vertex group.   AIRPOSE_HIP_LIB=$PWD/airpose_amd/libairpose_hip_ttrace.so python tools/probes/lbs_trace.py [bodies]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from airpose_amd import _native as N
from airpose_amd import smplx, smplx_model
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
body = smplx.SMPLX(model_data=smplx_model.make_synthetic_model(4321))
g = torch.Generator().manual_seed(n)
pose = torch.randn(n, 135, generator=g).to(dev)
betas = (torch.randn(n, 10, generator=g) * 0.5).to(dev)
cc = torch.tensor([960.0, 540.0]).expand(n, 2).contiguous().to(dev)
for _ in range(3):
    body.forward_fused(pose, betas, cc)
buf = torch.zeros(160, dtype=torch.int64, device=dev)
names = ["K step %d" % k for k in range(7)] + ["body %d" % k for k in range(8)] + ["fragment set copy"]
L = N.lib()
for rep in range(3):
    buf.zero_()
    L.ap_debug_set_trace(ctypes.c_void_p(buf.data_ptr())); body.forward_fused(pose, betas, cc); torch.cuda.synchronize(); L.ap_debug_set_trace(None)
    b = buf.cpu()
    for wg in (0, 1):
        t = [int(v) for v in b[wg * 24: wg * 24 + 24]]
        print("workgroup %d, second group of wave 0: %d cycles" % (100 * wg, t[16] - t[0]), " ".join("%s %d |" % (nm.replace("K step ", "k").replace("body ", "b"), t[i + 1] - t[i]) for i, nm in enumerate(names)))
