#!/usr/bin/env python
"""Cycle stamps of the second-cut bottleneck kernel (make trace: -DAP_TRACE build, libairpose_hip_trace.so): waves 0 and 4 of
workgroup 0, sixth tile.   AIRPOSE_HIP_LIB=$PWD/airpose_amd/libairpose_hip_trace.so python tools/probes/bneck2_trace.py [ds]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from airpose_amd import _native as N
dev = torch.device("cuda", 0)
L = N.lib()
bf = torch.bfloat16
p = lambda t: ctypes.c_void_p(t.data_ptr())
n, H = 512, 56
ds = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cin, k3 = (64, 128) if ds else (256, 64)
x = torch.randn(n, H, H, cin, device=dev).to(bf)
w1 = (torch.randn(128, cin, device=dev) * 0.05).to(bf)
w2 = (torch.randn(128, 576, device=dev) * 0.05).to(bf)
w3 = (torch.randn(256, k3, device=dev) * 0.05).to(bf)
sc = [torch.ones(c, device=dev) for c in (128, 128, 256)]
sh = [torch.zeros(c, device=dev) for c in (128, 128, 256)]
y = torch.empty(n, H, H, 256, device=dev, dtype=bf)
buf = torch.zeros(64, dtype=torch.int64, device=dev)
def run():
    N.check(L.ap_bottleneck64_nhwc(N.PRECISIONS["bf16"], p(x), p(w1), p(sc[0]), p(sh[0]), p(w2), p(sc[1]), p(sh[1]), p(w3), p(sc[2]), p(sh[2]),
                                   p(y), n, H, H, cin, ds, N.stream_ptr(dev)), "bneck")
for _ in range(3): run()
names = ["conv1", "barrier A", "epi1 + t1 write", "barrier B", "conv2", "epi2", "wait W3 + barrier C", "chunk 0", "chunk 1",
         "wait + barrier D", "chunk 2", "chunk 3"]
for rep in range(3):
    buf.zero_()
    L.ap_debug_set_trace(p(buf)); run(); torch.cuda.synchronize(); L.ap_debug_set_trace(None)
    b = buf.cpu()
    for wv in (0, 1):
        t = b[wv * 24: wv * 24 + 17]
        d = [int(t[i + 1] - t[i]) for i in range(12)]
        print("ds=%d wave %d tile total %6d cycles (100 MHz ticks x clock ratio): " % (ds, wv * 4, int(t[12] - t[0])) + "  ".join("%s=%d" % (names[i], d[i]) for i in range(12)))
        print("      chunk 1 in detail: fragment reads + 16 MFMAs issued=%d  tables of pair 0=%d  pair 0 arithmetic + 2 stores=%d  tables of pair 1=%d  pair 1 arithmetic + 2 stores=%d" % (
            int(t[13] - t[8]), int(t[14] - t[13]), int(t[15] - t[14]), int(t[16] - t[15]), int(t[9] - t[16])))
