#!/usr/bin/env python
"""Per-phase cycle stamps of the slab conv kernel (AP_TRACE build; workgroup 0, waves 0 and 7, taps 0..7 of chunk 1).
   make -C airpose_amd/csrc trace; AIRPOSE_HIP_LIB=$PWD/airpose_amd/libairpose_hip_trace.so python tools/probes/slab_trace.py"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from airpose_amd import _native as N
dev = torch.device("cuda", 0)
L = N.lib()
n, H, cin, cout = [int(v) for v in os.environ.get('SHAPE', '512,14,256,256').split(',')]
x = torch.randn(n, H, H, cin, device=dev).bfloat16()
cp = (cout + 127) // 128 * 128
w = (torch.randn(cp, 3, 3, cin, device=dev) * 0.02).bfloat16()
sc, sh = torch.ones(cp, device=dev), torch.zeros(cp, device=dev)
y = torch.empty(n, H, H, cout, device=dev, dtype=torch.bfloat16)
buf = torch.zeros(176, dtype=torch.int64, device=dev)
p = lambda t: ctypes.c_void_p(t.data_ptr())
L.ap_set_conv_config(14)
def run():
    N.check(L.ap_conv2d_nhwc(1, p(x), p(w), p(sc), p(sh), None, p(y), n, H, H, cin, cout, 3, 1, 1, 1, N.stream_ptr(dev)), "conv")
for _ in range(3): run()
L.ap_debug_set_trace(p(buf)); run(); torch.cuda.synchronize(); L.ap_debug_set_trace(None)
t = buf.cpu()[:160].view(2, 8, 10)
names = ["issue F1 reads", "wait F0", "cluster0+addr", "wait F1", "vmcnt", "barrier", "issue F0' reads", "cluster1+DMA"]
for wv, nm in ((0, "wave 0 (weights)"), (1, "wave 7 (slab)")):
    print(nm)
    for tap in range(8):
        d = [int(t[wv, tap, i + 1] - t[wv, tap, i]) for i in range(8)]
        nxt = int(t[wv, tap + 1, 0] - t[wv, tap, 0]) if tap < 7 else 0
        print("  tap %d: " % tap + " ".join("%s=%d" % (names[i], d[i]) for i in range(8)), "step=%d" % nxt)
