#!/usr/bin/env python
"""Micro-benchmark of the fused conv primitive (ap_conv2d_nhwc) on the ResNet-50 layer shapes.

  python tools/conv_bench.py --images 256 --cfgs -1,0,1,100 [--only l3.c2] [--iters 30] [--precision bf16]

Prints per shape and tile configuration: time (HIP events on the launch stream), TFLOP/s and the
minimal-traffic GB/s.  Data are random (not zero-filled): zero operands clock higher and flatter MFMAs."""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from airpose_amd import _native as N  # noqa: E402


def shapes(n):
    out, inpl, H = [], 64, 56
    for li, (pl, nb) in enumerate(zip((64, 128, 256, 512), (3, 4, 6, 3))):
        for bi in (0, 1):
            st = 2 if (bi == 0 and li > 0) else 1
            Ho = H // st
            tag = "l%d.%d" % (li + 1, bi)
            out.append((tag + ".c1", n, H, inpl, pl, 1, 1, 0, True, False))
            out.append((tag + ".c2", n, H, pl, pl, 3, st, 1, True, False))
            if bi == 0:
                out.append((tag + ".ds", n, H, inpl, pl * 4, 1, st, 0, False, False))
            out.append((tag + ".c3", n, Ho, pl, pl * 4, 1, 1, 0, True, True))
            inpl, H = pl * 4, Ho
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=256)
    ap.add_argument("--cfgs", default="-1")
    ap.add_argument("--only", default="")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--precision", default="bf16")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    L = N.lib()
    tdt = torch.bfloat16 if args.precision == "bf16" else torch.float32
    es = 2 if args.precision == "bf16" else 4
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    cfgs = [int(c) for c in args.cfgs.split(",")]
    tot = {c: 0.0 for c in cfgs}
    print("%-9s %8s %5s %5s | %s" % ("layer", "M", "N", "K", " | ".join("cfg %4d: us   TF/s  GB/s" % c for c in cfgs)))
    for (tag, n, H, cin, cout, k, st, pad, relu, res) in shapes(args.images):
        if args.only and args.only not in tag:
            continue
        Ho = (H + 2 * pad - k) // st + 1
        M = n * Ho * Ho
        x = torch.randn(n, H, H, cin, device=dev).to(tdt)
        cpad = (cout + 127) // 128 * 128
        w = (torch.randn(cpad, k, k, cin, device=dev) * (2.0 / (k * k * cin)) ** 0.5).to(tdt)
        sc, sh = torch.rand(cpad, device=dev) + 0.5, torch.randn(cpad, device=dev) * 0.1
        r = torch.randn(n, Ho, Ho, cout, device=dev).to(tdt) if res else None
        y = torch.empty(n, Ho, Ho, cout, device=dev, dtype=tdt)
        flops = 2.0 * M * cout * k * k * cin
        byts = (x.numel() / (st * st if k == 1 else 1) + y.numel() * (2 if res else 1)) * es
        cells = []
        for c in cfgs:
            L.ap_set_conv_config(c)

            def run():
                N.check(L.ap_conv2d_nhwc(N.PRECISIONS[args.precision], p(x), p(w), p(sc), p(sh), p(r), p(y), n, H, H,
                                         cin, cout, k, st, pad, int(relu), N.stream_ptr(dev)), "conv")
            try:
                for _ in range(3):
                    run()
            except RuntimeError:                           # configuration refuses this shape (e.g. conv_phase + residual)
                cells.append("%9s %6s %5s" % ("n/a", "", ""))
                tot[c] += float("nan")
                continue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / args.iters
            tot[c] += us
            cells.append("%9.1f %6.0f %5.0f" % (us, flops / us / 1e6, byts / us / 1e3))
        L.ap_set_conv_config(-1)
        print("%-9s %8d %5d %5d | %s" % (tag, M, cout, k * k * cin, " | ".join(cells)))
    print("sum us (one of each listed shape): " + "  ".join("cfg %d: %.0f" % (c, tot[c]) for c in cfgs))


if __name__ == "__main__":
    main()
