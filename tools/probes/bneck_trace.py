#!/usr/bin/env python
"""Cycle stamps of the fused bottleneck kernel (-DAP_TRACE build): wave 0 of workgroups 0 and 4096."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from airpose_amd import _native as N
dev = torch.device("cuda", 0)
L = N.lib()
bf = torch.bfloat16
p = lambda t: ctypes.c_void_p(t.data_ptr())
n, H = 512, 56
for ds in (0,):   # (only the identity kernel carries stamps)
    cin, k3 = (64, 128) if ds else (256, 64)
    x = torch.randn(n, H, H, cin, device=dev).to(bf)
    w1 = (torch.randn(128, cin, device=dev) * 0.05).to(bf)
    w2 = (torch.randn(128, 576, device=dev) * 0.05).to(bf)
    w3 = (torch.randn(256, k3, device=dev) * 0.05).to(bf)
    sc = [torch.ones(c, device=dev) for c in (128, 128, 256)]
    sh = [torch.zeros(c, device=dev) for c in (128, 128, 256)]
    y = torch.empty(n, H, H, 256, device=dev, dtype=bf)
    buf = torch.zeros(176, dtype=torch.int64, device=dev)
    def run():
        N.check(L.ap_bottleneck64_nhwc(p(x), p(w1), p(sc[0]), p(sh[0]), p(w2), p(sc[1]), p(sh[1]), p(w3), p(sc[2]), p(sh[2]),
                                       p(y), n, H, H, cin, ds, N.stream_ptr(dev)), "bneck")
    for _ in range(3): run()
    L.ap_debug_set_trace(p(buf)); run(); torch.cuda.synchronize(); L.ap_debug_set_trace(None)
    b = buf.cpu()
    names = ["setup+issue"] + ["p1 step %d" % i for i in range(1 if ds else 4)] + ["epi1"] + ["tap %d" % i for i in range(9)] + ["epi2"] + ["pass %d" % i for i in range(4)]
    if not ds:   # persistent kernel: third tile of workgroup 0, a weight wave (0) and an x wave (4)
        names = ["top+wait0", "kc0", "kc1", "kc2", "kc3", "prefetch+epi1"] + ["tap %d" % i for i in range(9)] + ["epi2"] + ["pass %d" % i for i in range(4)]
    for wg in (0, 1):
        t = b[wg * 40: wg * 40 + len(names) + 1]
        d = [int(t[i + 1] - t[i]) for i in range(len(names))]
        if not ds:
            f = b[wg * 40 + 26: wg * 40 + 32]
            print("      pass 1 detail: mfma step=%d epilogue=%d stage barrier=%d stage read=%d store issue=%d" % tuple(int(f[i + 1] - f[i]) for i in range(5)))
        print("ds=%d slot %d total %6d cycles: " % (ds, wg, int(t[len(names)] - t[0])) + "  ".join("%s=%d" % (names[i], d[i]) for i in range(len(names))))
