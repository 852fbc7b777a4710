#!/usr/bin/env python
"""Cycle stamps of the persistent stem + pool kernel (stem.hip built with -DAP_TRACE: tools/build_variant.sh stem trace -DAP_TRACE):
workgroup 3, strips 8-11 of its range; wave 0 (compute) and wave 8 (feed).
AIRPOSE_HIP_LIB=$PWD/airpose_amd/libairpose_hip_trace.so python tools/probes/stem_trace.py [images]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from airpose_amd import _native as N
from airpose_amd import copenet_model, weights as W
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
MEAN = os.path.join(R, "airpose_amd", "data", "smpl_mean_params.npz")
net = copenet_model.getcopenet(MEAN, precision=os.environ.get("PREC", "f16")).eval()
net.load_state_dict(W.to_torch(W.copenet_state_dict(7, MEAN)))
x = torch.randn(n, 3, 224, 224, generator=torch.Generator().manual_seed(1)).to(dev)
for _ in range(2):
    net.forward_feat_ext(x)
buf = torch.zeros(160, dtype=torch.int64, device=dev)
L = N.lib()
for rep in range(3):
    buf.zero_()
    L.ap_debug_set_trace(ctypes.c_void_p(buf.data_ptr())); net.forward_feat_ext(x); torch.cuda.synchronize(); L.ap_debug_set_trace(None)
    b = [int(v) for v in buf.cpu()]
    t0 = b[0]
    for it in range(4):
        c = b[it * 8: it * 8 + 4]; f = b[64 + it * 8: 64 + it * 8 + 5]
        print("strip %d  compute wave 0: start %6d | K loop %5d | epilogue %5d | barrier %5d      feed wave 8: start %6d | wait loads %5d | fill %5d | issue %5d | barrier %5d" % (
            it + 8, c[0] - t0, c[1] - c[0], c[2] - c[1], c[3] - c[2], f[0] - t0, f[1] - f[0], f[2] - f[1], f[3] - f[2], f[4] - f[3]))
    print()
