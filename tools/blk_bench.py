#!/usr/bin/env python3
"""Stand-alone timing of the image-resident layer3 bottleneck (ap_block_img_nhwc) against the kernels it replaces.
   python tools/blk_bench.py [--images 512] [--iters 20] [--precisions bf16,f16]
Per launch at 512 images: 223.6 GFLOP algorithmic, 615 MB of HBM traffic (x twice, out once)."""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from airpose_amd import _native as Nn  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=512)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--precisions", default="f16")
    ap.add_argument("--ref", type=int, default=1, help="also time conv1 (ring) + conv2 (slab) + conv3+identity (lean / ring) stand-alone")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    L = Nn.lib()
    for prec in a.precisions.split(","):
        bf = {"bf16": torch.bfloat16, "f16": torch.float16}[prec]
        g = torch.Generator().manual_seed(1)
        N, H = a.images, 14
        x = torch.randn(N, H, H, 1024, generator=g).to(bf).to(dev)
        w1 = (torch.randn(256, 1024, generator=g) * (2.0 / 1024) ** 0.5).to(bf).to(dev)
        w2 = (torch.randn(256, 2304, generator=g) * (2.0 / 2304) ** 0.5).to(bf).to(dev)
        w3 = (torch.randn(1024, 256, generator=g) * (2.0 / 256) ** 0.5).to(bf).to(dev)
        sc = [(torch.rand(c, generator=g) * 0.5 + 0.25).to(dev) for c in (256, 256, 1024)]
        sh = [(torch.randn(c, generator=g) * 0.1).to(dev) for c in (256, 256, 1024)]
        p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        st = Nn.stream_ptr(dev)
        B = Nn.PRECISIONS[prec]
        ws = torch.empty(L.ap_block_img_stream_bytes(), dtype=torch.uint8, device=dev)
        Nn.check(L.ap_block_img_pack(B, p(w1), p(w2), p(w3), p(ws), st), "pack")
        y = torch.empty_like(x)
        t1 = torch.empty(N, H, H, 256, dtype=bf, device=dev)
        t2 = torch.empty_like(t1)
        flops = 2.0 * N * 196 * (1024 * 256 + 2304 * 256 + 256 * 1024)

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
            print("%-44s %s images %d  %8.1f us  %7.0f TFLOP/s" % (name, prec, N, us, flops / us * 1e-6))
            return us
        timeit("block_img (one kernel)", lambda: Nn.check(L.ap_block_img_nhwc(B, p(x), p(ws), p(sc[0]), p(sh[0]), p(sc[1]), p(sh[1]), p(sc[2]),
                                                                               p(sh[2]), p(y), N, st), "blk"))
        if a.ref:
            def three():
                Nn.check(L.ap_conv2d_nhwc(B, p(x), p(w1), p(sc[0]), p(sh[0]), None, p(t1), N, H, H, 1024, 256, 1, 1, 0, 1, st), "c1")
                Nn.check(L.ap_conv2d_nhwc(B, p(t1), p(w2), p(sc[1]), p(sh[1]), None, p(t2), N, H, H, 256, 256, 3, 1, 1, 1, st), "c2")
                Nn.check(L.ap_conv2d_nhwc(B, p(t2), p(w3), p(sc[2]), p(sh[2]), p(x), p(y), N, H, H, 256, 1024, 1, 1, 0, 1, st), "c3")
            timeit("conv1 + conv2 + conv3 (automatic kernels)", three)


if __name__ == "__main__":
    main()
