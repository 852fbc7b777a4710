#!/usr/bin/env python3
"""Stand-alone timing of layer2.0's stride-2 3x3 (128 -> 128, 56 x 56 -> 28 x 28): conv_s2p.hip (polyphase, a quarter image per
workgroup) against the stand-alone convolution (ap_conv2d_nhwc: the ring kernel).  118.4 GFLOP per launch at 512 images.
   python tools/s2p_bench.py [--images 512,256] [--iters 30]"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from airpose_amd import _native as Nn  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", default="512,256")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--precision", default="f16")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    L = Nn.lib()
    B = Nn.PRECISIONS[a.precision]
    dt = {"bf16": torch.bfloat16, "f16": torch.float16}[a.precision]
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    st = Nn.stream_ptr(dev)
    for N in [int(v) for v in a.images.split(",")]:
        g = torch.Generator().manual_seed(N)
        x = torch.randn(N, 56, 56, 128, generator=g).to(dt).to(dev)
        w = (torch.randn(128, 3, 3, 128, generator=g) * (2.0 / 1152) ** 0.5).to(dt).to(dev)
        sc = (torch.rand(128, generator=g) + 0.5).to(dev)
        sh = (torch.randn(128, generator=g) * 0.1).to(dev)
        y = torch.empty(N, 28, 28, 128, dtype=dt, device=dev)
        ws = torch.empty(L.ap_conv_s2p_stream_bytes(), dtype=torch.uint8, device=dev)
        Nn.check(L.ap_conv_s2p_pack(B, p(w), p(ws), st), "pack")
        flops = 2.0 * N * 784 * 128 * 1152
        runs = [("conv_s2p (polyphase, quarter image in LDS)", lambda: L.ap_conv_s2p_nhwc(B, p(x), p(ws), p(sc), p(sh), p(y), N, 0, st)),
                ("conv_s2p, tiled output", lambda: L.ap_conv_s2p_nhwc(B, p(x), p(ws), p(sc), p(sh), p(y), N, 1, st)),
                ("ring kernel (ap_conv2d_nhwc)", lambda: L.ap_conv2d_nhwc(B, p(x), p(w), p(sc), p(sh), None, p(y), N, 56, 56, 128, 128, 3, 2, 1, 1, st))]
        for name, fn in runs:
            for _ in range(3):
                Nn.check(fn(), name)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / a.iters
            print("%-46s %s images %4d  %8.1f us  %7.0f TFLOP/s" % (name, a.precision, N, us, flops / us * 1e-6))


if __name__ == "__main__":
    main()
