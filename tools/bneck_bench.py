#!/usr/bin/env python3
"""Stand-alone timing of the fused layer1 bottleneck (ap_bottleneck64_nhwc), both 16-bit storage types.
   python tools/bneck_bench.py [--images 512] [--iters 20] [--precisions bf16,f16]
Per launch: 0.2235 TFLOP and 1.644 GB of algorithmic HBM traffic at 512 images (x in + out)."""
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
    ap.add_argument("--precisions", default="bf16,f16")
    ap.add_argument("--ds", type=int, default=0, help="1: the first block of layer1 (cin 64, folded downsample)")
    ap.add_argument("--tail", type=int, default=0, help="1: also time the tail variant (+ conv1 of layer2.0; y_even 0 / 1) and the stand-alone conv1")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    L = Nn.lib()
    for prec in a.precisions.split(","):
        bf = {"bf16": torch.bfloat16, "f16": torch.float16}[prec]
        g = torch.Generator().manual_seed(1)
        N, H = a.images, 56
        cin = 64 if a.ds else 256
        x = torch.randn(N, H, H, cin, generator=g).to(bf).to(dev)
        w1 = (torch.randn(128, cin, generator=g) * (2.0 / cin) ** 0.5).to(bf).to(dev)
        w2 = (torch.randn(128, 576, generator=g) * (2.0 / 576) ** 0.5).to(bf).to(dev)
        w3 = (torch.randn(256, 128 if a.ds else 64, generator=g) * (2.0 / 64) ** 0.5).to(bf).to(dev)
        sc = [(torch.rand(c, generator=g) + 0.5).to(dev) for c in (128, 128, 256)]
        sh = [(torch.randn(c, generator=g) * 0.1).to(dev) for c in (128, 128, 256)]
        y = torch.empty(N, H, H, 256, dtype=bf, device=dev)
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        st = Nn.stream_ptr(dev)
        flops = 2.0 * N * H * H * (cin * 64 + 576 * 64 + (128 if a.ds else 64) * 256)
        byts = 2.0 * N * H * H * (cin + 256)
        if True:
            call = lambda: Nn.check(L.ap_bottleneck64_nhwc(Nn.PRECISIONS[prec], p(x), p(w1), p(sc[0]), p(sh[0]), p(w2), p(sc[1]), p(sh[1]), p(w3),
                                                           p(sc[2]), p(sh[2]), p(y), N, H, H, cin, int(a.ds), st), "bneck")
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
            print("ds %d  %s  images %d  %.1f us  %.0f TFLOP/s  %.2f TB/s (algorithmic)" % (a.ds, prec, N, us, flops / us * 1e-6, byts / us * 1e-6))
        if a.tail and not a.ds:
            w1n = (torch.randn(128, 256, generator=g) * (2.0 / 256) ** 0.5).to(bf).to(dev)
            s1n, h1n = (torch.rand(128, generator=g) + 0.5).to(dev), (torch.randn(128, generator=g) * 0.1).to(dev)
            t1n = torch.empty(N, H, H, 128, dtype=bf, device=dev)
            calls = {
                "tail, y in full": lambda: Nn.check(L.ap_bottleneck64_tail_nhwc(Nn.PRECISIONS[prec], p(x), p(w1), p(sc[0]), p(sh[0]), p(w2), p(sc[1]), p(sh[1]),
                    p(w3), p(sc[2]), p(sh[2]), p(y), p(w1n), p(s1n), p(h1n), p(t1n), 0, N, H, H, st), "tail"),
                "tail, y even pixels": lambda: Nn.check(L.ap_bottleneck64_tail_nhwc(Nn.PRECISIONS[prec], p(x), p(w1), p(sc[0]), p(sh[0]), p(w2), p(sc[1]), p(sh[1]),
                    p(w3), p(sc[2]), p(sh[2]), p(y), p(w1n), p(s1n), p(h1n), p(t1n), 1, N, H, H, st), "tail"),
                "stand-alone conv1 256 -> 128": lambda: Nn.check(L.ap_conv2d_nhwc(Nn.PRECISIONS[prec], p(y), p(w1n), p(s1n), p(h1n), None, p(t1n), N, H, H, 256, 128,
                    1, 1, 0, 1, st), "c1n"),
            }
            for name, call in calls.items():
                for _ in range(3):
                    call()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    call()
                e1.record()
                torch.cuda.synchronize()
                print("   %-30s %s  %.1f us" % (name, prec, e0.elapsed_time(e1) * 1e3 / a.iters))


if __name__ == "__main__":
    main()
