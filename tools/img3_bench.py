#!/usr/bin/env python3
"""Stand-alone timing of layer2's 3x3 (128 -> 128 at 28 x 28): conv_img3.hip (half an image resident in LDS) against the slab kernel.
   python tools/img3_bench.py [--images 512] [--iters 20]
Per launch at 512 images: 118.4 GFLOP, 103 MB in + 103 MB out."""
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
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--precisions", default="f16")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    L = Nn.lib()
    for prec in a.precisions.split(","):
        bf = {"bf16": torch.bfloat16, "f16": torch.float16}[prec]
        B = Nn.PRECISIONS[prec]
        for N in [int(v) for v in a.images.split(",")]:
            g = torch.Generator().manual_seed(1)
            x = torch.randn(N, 28, 28, 128, generator=g).to(bf).to(dev)
            w = (torch.randn(128, 3, 3, 128, generator=g) * (2.0 / 1152) ** 0.5).to(bf).to(dev)
            sc, sh = (torch.rand(128, generator=g) + 0.5).to(dev), (torch.randn(128, generator=g) * 0.1).to(dev)
            y = torch.empty_like(x)
            p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
            st = Nn.stream_ptr(dev)
            ws = torch.empty(L.ap_conv_img3_stream_bytes(), dtype=torch.uint8, device=dev)
            Nn.check(L.ap_conv_img3_pack(B, p(w), p(ws), st), "pack")
            flops = 2.0 * N * 784 * 128 * 1152

            def timeit(name, call):
                for _ in range(3):
                    call()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    call()
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / a.iters
                print("%-34s %s images %4d  %8.1f us  %7.0f TFLOP/s" % (name, prec, N, us, flops / us * 1e-6))
            timeit("conv_img3 (half image in LDS)", lambda: Nn.check(L.ap_conv_img3_nhwc(B, p(x), p(ws), p(sc), p(sh), p(y), N, 0, st), "img3"))
            timeit("conv_img3, tiled output", lambda: Nn.check(L.ap_conv_img3_nhwc(B, p(x), p(ws), p(sc), p(sh), p(y), N, 1, st), "img3"))
            timeit("slab kernel (ap_conv2d_nhwc)", lambda: Nn.check(L.ap_conv2d_nhwc(B, p(x), p(w), p(sc), p(sh), None, p(y), N, 28, 28, 128, 128, 3, 1, 1, 1, st), "slab"))


if __name__ == "__main__":
    main()
