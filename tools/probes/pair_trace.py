#!/usr/bin/env python
"""Cycle stamps of the pair kernel (make trace: -DAP_TRACE, libairpose_hip_trace.so): wave 0 of workgroups 0 and 300, chunk 1.
   AIRPOSE_HIP_LIB=$PWD/airpose_amd/libairpose_hip_trace.so python tools/probes/pair_trace.py [l2|l23|l3]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from airpose_amd import _native as N
dev = torch.device("cuda", 0)
L = N.lib()
p = lambda t: ctypes.c_void_p(t.data_ptr())
case = sys.argv[1] if len(sys.argv) > 1 else "l3"
H, P, N1 = {"l2": (28, 128, 128), "l23": (28, 128, 256), "l3": (14, 256, 256)}[case]
n = 256
M, C3 = n * H * H, 4 * P
t2 = torch.randn(M, P, device=dev).clamp_min(0).to(torch.bfloat16)
x = torch.randn(M, C3, device=dev).clamp_min(0).to(torch.bfloat16)
w3 = (torch.randn(C3, P, device=dev) * (2.0 / P) ** 0.5).to(torch.bfloat16)
w1 = (torch.randn(N1, C3, device=dev) * (2.0 / C3) ** 0.5).to(torch.bfloat16)
s3, h3 = torch.rand(C3, device=dev) + 0.5, torch.randn(C3, device=dev) * 0.1
s1, h1 = torch.rand(N1, device=dev) + 0.5, torch.randn(N1, device=dev) * 0.1
out, t1n = torch.empty(M, C3, dtype=torch.bfloat16, device=dev), torch.empty(M, N1, dtype=torch.bfloat16, device=dev)
buf = torch.zeros(160, dtype=torch.int64, device=dev)
BF = N.PRECISIONS["bf16"]
ws = torch.empty(L.ap_conv_pair_stream_bytes(P, 0, N1), dtype=torch.uint8, device=dev)
N.check(L.ap_conv_pair_pack(BF, p(w3), p(w1), P, 0, N1, p(ws), N.stream_ptr(dev)), "pack")
def run():
    N.check(L.ap_conv_pair_nhwc(BF, p(t2), p(ws), p(s3), p(h3), p(x), p(s1), p(h1), p(out), p(t1n), M, P, N1, N.stream_ptr(dev)), "pair")
for _ in range(3): run()
KP, SPC = P // 64, P // 64 + 2 * (N1 // 128)
for rep in range(2):
    buf.zero_()
    L.ap_debug_set_trace(p(buf)); run(); torch.cuda.synchronize(); L.ap_debug_set_trace(None)
    b = buf.cpu()
    for wg in (0, 1):
        t = [int(v) for v in b[wg * 32: wg * 32 + 32]]
        parts = []
        for j in range(SPC):
            nxt = t[3 * (j + 1)] if j + 1 < SPC else t[30]
            body = nxt - t[3 * j + 2]
            extra = ""
            if j == KP - 1:
                body = t[28] - t[3 * j + 2]
                extra = " epilogue=%d" % (t[29] - t[28])
            parts.append("step %d [%s]: vmcnt wait=%d barrier=%d body=%d%s" % (j, "conv3" if j < KP else "conv1", t[3 * j + 1] - t[3 * j], t[3 * j + 2] - t[3 * j + 1], body, extra))
        print("%s workgroup slot %d, chunk 1: total %d cycles\n   " % (case, wg, t[30] - t[0]) + "\n   ".join(parts))
