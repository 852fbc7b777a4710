#!/usr/bin/env python
"""Per-phase cycle stamps of the pipelined conv kernel (workgroup 0, waves 0 and 4, K steps 8..15)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from airpose_amd import _native as N
dev = torch.device("cuda", 0)
L = N.lib()
import os
n, H, cin, cout, k = [int(v) for v in os.environ.get('SHAPE', '512,14,256,256,3').split(',')]
RES = int(os.environ.get('RES', '0'))
x = torch.randn(n, H, H, cin, device=dev).bfloat16()
cp = (cout + 127) // 128 * 128
w = (torch.randn(cp, k, k, cin, device=dev) * 0.02).bfloat16()
sc, sh = torch.ones(cp, device=dev), torch.zeros(cp, device=dev)
res = torch.randn(n, H, H, cout, device=dev).bfloat16() if RES else None
y = torch.empty(n, H, H, cout, device=dev, dtype=torch.bfloat16)
buf = torch.zeros(176, dtype=torch.int64, device=dev)
p = lambda t: ctypes.c_void_p(t.data_ptr())
L.ap_set_conv_config(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
def run():
    N.check(L.ap_conv2d_nhwc(1, p(x), p(w), p(sc), p(sh), p(res) if RES else None, p(y), n, H, H, cin, cout, k, 1, (k - 1) // 2, 1, N.stream_ptr(dev)), "conv")
for _ in range(3): run()
L.ap_debug_set_trace(p(buf)); run(); torch.cuda.synchronize(); L.ap_debug_set_trace(None)
b = buf.cpu()[160:168]
print('block phases (cycles): setup=%d issue=%d first-wait=%d kloop=%d ldswrite=%d barrier=%d readback+store=%d total=%d' % tuple([int(b[i+1]-b[i]) for i in range(7)] + [int(b[7]-b[0])]))
t = buf.cpu()[:160].view(2, 8, 10)
names = ["top>F1 issue", "wait F0", "cluster0", "vmcnt", "barrier", "issue glds", "F0' issue", "wait F1", "cluster1", "->next top"]
for wv in (0, 1):
    print("wave", wv * 4)
    for kt in range(8):
        d = [int(t[wv, kt, i + 1] - t[wv, kt, i]) for i in range(9)]
        nxt = int(t[wv, kt + 1, 0] - t[wv, kt, 9]) if kt < 7 else 0
        print("  kt %2d: " % (kt + 8) + " ".join("%s=%d" % (names[i], d[i]) for i in range(9)), "loop=%d" % nxt, "total=%d" % (int(t[wv, kt, 9] - t[wv, kt, 0]) + nxt))
