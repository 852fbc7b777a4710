#!/usr/bin/env python3
"""Do an MFMA-bound CU-owning kernel and HBM-bound kernels overlap when they run on DISJOINT CUs of one chip?
Stream M: ap_block_img_nhwc on n_m images (one workgroup per image: n_m CUs, all of their LDS and registers).
Stream H: the HBM-bound layer2 tail of a block (1x1 conv 128 -> 512 + identity + ReLU, 28 x 28) on n_h images: small workgroups
that run wherever a CU is free.  Timed alone and together (wall time until both streams are done).
   python tools/probes/overlap_probe.py [--nm 128,171,192,256] [--nh 256]"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from airpose_amd import _native as Nn  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nm", default="128,171,192,256")
    ap.add_argument("--nh", type=int, default=256)
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    L = Nn.lib()
    bf, B = torch.float16, Nn.PRECISIONS["f16"]
    g = torch.Generator().manual_seed(1)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    # M: a layer3 identity block
    w1 = (torch.randn(256, 1024, generator=g) * (2.0 / 1024) ** 0.5).to(bf).to(dev)
    w2 = (torch.randn(256, 2304, generator=g) * (2.0 / 2304) ** 0.5).to(bf).to(dev)
    w3 = (torch.randn(1024, 256, generator=g) * (2.0 / 256) ** 0.5).to(bf).to(dev)
    sc = [(torch.rand(c, generator=g) * 0.5 + 0.25).to(dev) for c in (256, 256, 1024)]
    sh = [(torch.randn(c, generator=g) * 0.1).to(dev) for c in (256, 256, 1024)]
    ws = torch.empty(L.ap_block_img_stream_bytes(), dtype=torch.uint8, device=dev)
    Nn.check(L.ap_block_img_pack(B, p(w1), p(w2), p(w3), p(ws), Nn.stream_ptr(dev)), "pack")
    # H: conv3 + identity of a layer2 block (K = 128: 0.2 GFLOP per image against 1.2 MB of traffic)
    nh = a.nh
    t2 = torch.randn(nh, 28, 28, 128, generator=g).to(bf).to(dev)
    xh = torch.randn(nh, 28, 28, 512, generator=g).to(bf).to(dev)
    yh = torch.empty_like(xh)
    wh = (torch.randn(512, 128, generator=g) * (2.0 / 128) ** 0.5).to(bf).to(dev)
    sch, shh = (torch.rand(512, generator=g) * 0.5 + 0.25).to(dev), (torch.randn(512, generator=g) * 0.1).to(dev)
    sM, sH = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()

    def run_h(k):
        with torch.cuda.stream(sH):
            for _ in range(k):
                Nn.check(L.ap_conv2d_nhwc(B, p(t2), p(wh), p(sch), p(shh), p(xh), p(yh), nh, 28, 28, 128, 512, 1, 1, 0, 1,
                                          ctypes.c_void_p(sH.cuda_stream)), "h")

    def wall(fn):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        torch.cuda.current_stream().synchronize()
        import time
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e6

    for nm in [int(v) for v in a.nm.split(",")]:
        x = torch.randn(nm, 14, 14, 1024, generator=g).to(bf).to(dev)
        y = torch.empty_like(x)

        def run_m(k):
            with torch.cuda.stream(sM):
                for _ in range(k):
                    Nn.check(L.ap_block_img_nhwc(B, p(x), p(ws), p(sc[0]), p(sh[0]), p(sc[1]), p(sh[1]), p(sc[2]), p(sh[2]), p(y), nm,
                                                 ctypes.c_void_p(sM.cuda_stream)), "m")
        run_m(3); run_h(3)
        tm = min(wall(lambda: run_m(a.reps)) for _ in range(3)) / a.reps
        # as many H launches as fill the same time
        th1 = min(wall(lambda: run_h(a.reps)) for _ in range(3)) / a.reps
        kh = max(1, int(round(tm / th1)))
        th = min(wall(lambda: run_h(a.reps * kh)) for _ in range(3)) / a.reps
        both = min(wall(lambda: (run_m(a.reps), run_h(a.reps * kh))) for _ in range(3)) / a.reps
        print("blk_img %3d images: alone %6.1f us | H (%d images x %d launches): alone %6.1f us | together %6.1f us  (sum %6.1f, max %6.1f)"
              % (nm, tm, nh, kh, th, both, tm + th, max(tm, th)))


if __name__ == "__main__":
    main()
