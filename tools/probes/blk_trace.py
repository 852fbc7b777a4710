#!/usr/bin/env python
"""Cycle stamps of the image-resident layer3 block (make trace: -DAP_TRACE): wave 0 of workgroups 0 and 100, their second image.
   AIRPOSE_HIP_LIB=$PWD/airpose_amd/libairpose_hip_trace.so python tools/probes/blk_trace.py"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from airpose_amd import _native as N
dev = torch.device("cuda", 0)
L = N.lib()
p = lambda t: ctypes.c_void_p(t.data_ptr())
n, H = 512, 14
bf = torch.float16
x = torch.randn(n, H, H, 1024, device=dev).clamp_min(0).to(bf)
w1 = (torch.randn(256, 1024, device=dev) * (2.0 / 1024) ** 0.5).to(bf)
w2 = (torch.randn(256, 2304, device=dev) * (2.0 / 2304) ** 0.5).to(bf)
w3 = (torch.randn(1024, 256, device=dev) * (2.0 / 256) ** 0.5).to(bf)
sc = [torch.rand(c, device=dev) * 0.5 + 0.25 for c in (256, 256, 1024)]
sh = [torch.randn(c, device=dev) * 0.1 for c in (256, 256, 1024)]
y = torch.empty_like(x)
B = N.PRECISIONS["f16"]
ws = torch.empty(L.ap_block_img_stream_bytes(), dtype=torch.uint8, device=dev)
N.check(L.ap_block_img_pack(B, p(w1), p(w2), p(w3), p(ws), N.stream_ptr(dev)), "pack")
run = lambda: N.check(L.ap_block_img_nhwc(B, p(x), p(ws), p(sc[0]), p(sh[0]), p(sc[1]), p(sh[1]), p(sc[2]), p(sh[2]), p(y), n, N.stream_ptr(dev)), "blk")
for _ in range(3): run()
buf = torch.zeros(160, dtype=torch.int64, device=dev)
names = ["prologue (x chunks 0, 1 from HBM, 2 barriers)", "conv1 chunks 0-1", "conv1 chunks 2-15", "t1 -> LDS, zero, barrier", "conv2 (72 K steps)",
         "settle, barrier, t2 -> LDS, barrier", "conv3 chunk 0", "conv3 chunk 1: K loop", "   chunk 1: wait identity", "   chunk 1: epilogue + stores", "conv3 chunks 2-7"]
for rep in range(2):
    buf.zero_()
    L.ap_debug_set_trace(p(buf)); run(); torch.cuda.synchronize(); L.ap_debug_set_trace(None)
    b = buf.cpu()
    for wg in (0, 1):
        t = [int(v) for v in b[wg * 24: wg * 24 + 24]]
        print("workgroup %d, second image: %d cycles" % (100 * wg, t[11] - t[0]))
        for i, nm in enumerate(names):
            print("   %-50s %7d" % (nm, t[i + 1] - t[i]))
