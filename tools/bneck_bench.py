#!/usr/bin/env python
"""Micro-benchmark of the fused layer1 bottleneck (ap_bottleneck64_nhwc) against its separate convolutions.

  python tools/bneck_bench.py --images 512 [--iters 20]

Prints time, algorithmic HBM GB/s (block input + output once) and TFLOP/s per variant, and a checksum of the
output (for A/B runs of differently built libraries: AIRPOSE_HIP_LIB=...)."""
import argparse
import ctypes
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from airpose_amd import _native as N  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=512)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--hw", type=int, default=56)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    L = N.lib()
    bf = torch.bfloat16
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    n, H = args.images, args.hw
    g = torch.Generator(device="cpu").manual_seed(3)
    torch.manual_seed(7)
    for ds in (0, 1):
        cin = 64 if ds else 256
        x = torch.randn(n, H, H, cin, device=dev).to(bf)
        w1 = (torch.randn(128, cin, generator=g) * (2.0 / cin) ** 0.5).to(bf).to(dev)
        w2 = (torch.randn(128, 576, generator=g) * (2.0 / 576) ** 0.5).to(bf).to(dev)
        k3 = 128 if ds else 64
        w3 = (torch.randn(256, k3, generator=g) * (2.0 / k3) ** 0.5).to(bf).to(dev)
        sc = [(torch.rand(c, generator=g) + 0.5).to(dev) for c in (128, 128, 256)]
        sh = [(torch.randn(c, generator=g) * 0.1).to(dev) for c in (128, 128, 256)]
        y = torch.empty(n, H, H, 256, device=dev, dtype=bf)
        t1 = torch.empty(n, H, H, 64, device=dev, dtype=bf)
        t2 = torch.empty(n, H, H, 64, device=dev, dtype=bf)
        y2 = torch.empty(n, H, H, 256, device=dev, dtype=bf)

        def fused():
            N.check(L.ap_bottleneck64_nhwc(p(x), p(w1), p(sc[0]), p(sh[0]), p(w2), p(sc[1]), p(sh[1]), p(w3), p(sc[2]),
                                           p(sh[2]), p(y), n, H, H, cin, ds, N.stream_ptr(dev)), "bneck")

        def separate():         # (identity variant only: conv1, conv2, conv3 + residual)
            B = N.PRECISIONS["bf16"]
            N.check(L.ap_conv2d_nhwc(B, p(x), p(w1), p(sc[0]), p(sh[0]), None, p(t1), n, H, H, cin, 64, 1, 1, 0, 1,
                                     N.stream_ptr(dev)), "c1")
            N.check(L.ap_conv2d_nhwc(B, p(t1), p(w2), p(sc[1]), p(sh[1]), None, p(t2), n, H, H, 64, 64, 3, 1, 1, 1,
                                     N.stream_ptr(dev)), "c2")
            N.check(L.ap_conv2d_nhwc(B, p(t2), p(w3), p(sc[2]), p(sh[2]), p(x), p(y2), n, H, H, 64, 256, 1, 1, 0, 1,
                                     N.stream_ptr(dev)), "c3")

        def timeit(fn):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / args.iters

        px = n * H * H
        flops = 2.0 * px * (cin * 64 + 576 * 64 + k3 * 256)
        byts = px * (cin + 256) * 2
        us = timeit(fused)
        digest = hashlib.sha1(y.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:12]
        line = "ds=%d fused %8.1f us  %6.0f GB/s  %6.0f TF/s  sha1 %s" % (ds, us, byts / us / 1e3, flops / us / 1e6, digest)
        if not ds:
            us2 = timeit(separate)
            err = (y.float() - y2.float()).abs().max().item() / y2.float().abs().max().item()
            line += "   | separate %8.1f us  (max rel diff %.2e)" % (us2, err)
        print(line)


if __name__ == "__main__":
    main()
