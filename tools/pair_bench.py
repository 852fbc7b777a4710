#!/usr/bin/env python
"""A/B of the fused conv3 -> conv1 pair kernel (conv_pair.hip) against the two stand-alone launches the automatic choice
makes for the same layers, interleaved rounds in ONE process (HIP events on the launch stream, random non-zero data).

  python tools/pair_bench.py --images 256 [--iters 20] [--rounds 5]
"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from airpose_amd import _native as N  # noqa: E402

CASES = [("l2 c3->c1", 28, 128, 128), ("l2.3->l3.0", 28, 128, 256), ("l3 c3->c1", 14, 256, 256)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=256)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--nocheck", action="store_true", help="timing-only builds (PR_ABLATE) give wrong results")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    L = N.lib()
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    bf = N.PRECISIONS["bf16"]
    for tag, H, P, N1 in CASES:
        n = args.images
        M, C3 = n * H * H, 4 * P
        t2 = torch.randn(M, P, device=dev).clamp_min(0).to(torch.bfloat16)
        x = torch.randn(M, C3, device=dev).clamp_min(0).to(torch.bfloat16)
        w3 = (torch.randn(C3, P, device=dev) * (2.0 / P) ** 0.5).to(torch.bfloat16)
        w1 = (torch.randn(N1, C3, device=dev) * (2.0 / C3) ** 0.5).to(torch.bfloat16)
        s3, h3 = torch.rand(C3, device=dev) + 0.5, torch.randn(C3, device=dev) * 0.1
        s1, h1 = torch.rand(N1, device=dev) + 0.5, torch.randn(N1, device=dev) * 0.1
        out, t1n = torch.empty(M, C3, dtype=torch.bfloat16, device=dev), torch.empty(M, N1, dtype=torch.bfloat16, device=dev)
        out2, t1b = torch.empty_like(out), torch.empty_like(t1n)
        st = N.stream_ptr(dev)
        ws = torch.empty(L.ap_conv_pair_stream_bytes(P, 0, N1), dtype=torch.uint8, device=dev)   # caller-owned weight stream
        N.check(L.ap_conv_pair_pack(bf, p(w3), p(w1), P, 0, N1, p(ws), st), "pack")

        def fused():
            N.check(L.ap_conv_pair_nhwc(bf, p(t2), p(ws), p(s3), p(h3), p(x), p(s1), p(h1), p(out), p(t1n), M, P, N1, st), "pair")

        def two():
            N.check(L.ap_conv2d_nhwc(bf, p(t2), p(w3), p(s3), p(h3), p(x), p(out2), n, H, H, P, C3, 1, 1, 0, 1, st), "c3")
            N.check(L.ap_conv2d_nhwc(bf, p(out2), p(w1), p(s1), p(h1), None, p(t1b), n, H, H, C3, N1, 1, 1, 0, 1, st), "c1")

        res = {"fused": [], "two": []}
        for f in (fused, two):
            for _ in range(3):
                f()
        for _ in range(args.rounds):
            for nm, f in (("fused", fused), ("two", two)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    f()
                e1.record()
                torch.cuda.synchronize()
                res[nm].append(e0.elapsed_time(e1) * 1e3 / args.iters)
        assert args.nocheck or (torch.equal(out, out2) and torch.equal(t1n, t1b))
        flops = 2.0 * M * (C3 * P + N1 * C3)
        b_f = (M * P + 2 * M * C3 + M * N1) * 2                  # fused: t2 + identity in, out + t1' out
        b_t = b_f + M * C3 * 2                                   # two launches: conv1 reads `out` back
        med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
        print("%-11s M=%7d P=%3d N1=%3d | fused %7.1f us (min %7.1f) %6.0f TF/s %5.0f GB/s | two launches %7.1f us (min %7.1f) %6.0f TF/s "
              "%5.0f GB/s | x%.2f" % (tag, M, P, N1, med["fused"], min(res["fused"]), flops / med["fused"] / 1e6, b_f / med["fused"] / 1e3,
                                      med["two"], min(res["two"]), flops / med["two"] / 1e6, b_t / med["two"] / 1e3, med["two"] / med["fused"]))


if __name__ == "__main__":
    main()
