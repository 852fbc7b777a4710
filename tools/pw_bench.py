#!/usr/bin/env python3
"""Stand-alone timing of the one-wave-per-SIMD pointwise kernel (ap_conv_pw_nhwc) against the generic kernels on layer4's shapes.
   python tools/pw_bench.py [--images 512] [--iters 20] [--precision f16]"""
import argparse, ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from airpose_amd import _native as Nn  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=512)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--precision", default="f16")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    L = Nn.lib()
    bf = {"bf16": torch.bfloat16, "f16": torch.float16}[a.precision]
    B = Nn.PRECISIONS[a.precision]
    st = Nn.stream_ptr(dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    n = a.images
    for name, M, Cin, Cout, ident in (("l4.0.c1", n * 196, 1024, 512, False), ("l4.1.c1", n * 49, 2048, 512, False),
                                      ("l4.1.c3+id", n * 49, 512, 2048, True), ("l3.1.c1", n * 196, 1024, 256, False),
                                      ("l3.5.c3+id", n * 196, 256, 1024, True)):
        g = torch.Generator().manual_seed(1)
        x = torch.randn(M, Cin, generator=g).to(bf).to(dev)
        w = (torch.randn(Cout, Cin, generator=g) * (2.0 / Cin) ** 0.5).to(bf).to(dev)
        res = torch.randn(M, Cout, generator=g).to(bf).to(dev) if ident else None
        sc, sh = (torch.rand(Cout, generator=g) * 0.5 + 0.5).to(dev), (torch.randn(Cout, generator=g) * 0.1).to(dev)
        ws = torch.empty(L.ap_conv_pw_stream_bytes(Cin, Cout), dtype=torch.uint8, device=dev)
        Nn.check(L.ap_conv_pw_pack(B, p(w), Cin, Cout, p(ws), st), "pack")
        y, y2 = torch.empty(M, Cout, dtype=bf, device=dev), torch.empty(M, Cout, dtype=bf, device=dev)
        flops = 2.0 * M * Cin * Cout
        byts = (M * Cin + M * Cout * (2 if ident else 1)) * 2

        def timeit(call):
            for _ in range(3):
                call()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                call()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / a.iters
        t_pw = timeit(lambda: Nn.check(L.ap_conv_pw_nhwc(B, p(x), p(ws), p(sc), p(sh), p(res), p(y), M, Cin, Cout, st), "pw"))
        t_gen = timeit(lambda: Nn.check(L.ap_conv2d_nhwc(B, p(x), p(w), p(sc), p(sh), p(res), p(y2), M // 196, 14, 14, Cin, Cout, 1, 1, 0, 1, st), "conv"))
        print("%-11s M=%7d K=%4d N=%4d | conv_pw %7.1f us %6.0f TF/s %5.0f GB/s | generic %7.1f us %6.0f TF/s | x%.2f | equal %s" % (
            name, M, Cin, Cout, t_pw, flops / t_pw * 1e-6, byts / t_pw * 1e-3, t_gen, flops / t_gen * 1e-6, t_gen / t_pw, bool(torch.equal(y, y2))))


def k3_rows(a):
    """conv2 of a stage's first block (3 x 3, stride 2): nine pointwise taps on conv_pw against the generic kernel."""
    dev = torch.device("cuda", 0)
    L = Nn.lib()
    bf = {"bf16": torch.bfloat16, "f16": torch.float16}[a.precision]
    B = Nn.PRECISIONS[a.precision]
    st = Nn.stream_ptr(dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    for name, H, Cin, Cout in (("l4.0.c2", 14, 512, 512), ("l3.0.c2", 28, 256, 256)):
        n = a.images
        g = torch.Generator().manual_seed(3)
        x = torch.randn(n, H, H, Cin, generator=g).to(bf).to(dev)
        w = (torch.randn(Cout, 9 * Cin, generator=g) * (2.0 / (9 * Cin)) ** 0.5).to(bf).to(dev)
        sc, sh = (torch.rand(Cout, generator=g) * 0.5 + 0.5).to(dev), (torch.randn(Cout, generator=g) * 0.1).to(dev)
        ws = torch.empty(L.ap_conv_pw_stream_bytes(9 * Cin, Cout), dtype=torch.uint8, device=dev)
        Nn.check(L.ap_conv_pw_pack(B, p(w), 9 * Cin, Cout, p(ws), st), "pack")
        Ho = H // 2
        y, y2 = torch.empty(n, Ho, Ho, Cout, dtype=bf, device=dev), torch.empty(n, Ho, Ho, Cout, dtype=bf, device=dev)

        def timeit(call):
            for _ in range(3):
                call()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                call()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / a.iters
        t_pw = timeit(lambda: Nn.check(L.ap_conv_pw_k3s2_nhwc(B, p(x), p(ws), p(sc), p(sh), p(y), n, H, Cin, Cout, st), "pw k3"))
        t_gen = timeit(lambda: Nn.check(L.ap_conv2d_nhwc(B, p(x), p(w), p(sc), p(sh), None, p(y2), n, H, H, Cin, Cout, 3, 2, 1, 1, st), "conv"))
        flops = 2.0 * n * Ho * Ho * 9 * Cin * Cout
        print("%-11s M=%7d K=%4d N=%4d | conv_pw %7.1f us %6.0f TF/s | generic %7.1f us %6.0f TF/s | x%.2f | equal %s" % (
            name, n * Ho * Ho, 9 * Cin, Cout, t_pw, flops / t_pw * 1e-6, t_gen, flops / t_gen * 1e-6, t_gen / t_pw, bool(torch.equal(y, y2))))


def ds_rows(a):
    """conv3 + folded downsample of a stage's first block: conv_pw's two-segment form against the pair kernel (layer3.0's shape)."""
    dev = torch.device("cuda", 0)
    L = Nn.lib()
    bf = {"bf16": torch.bfloat16, "f16": torch.float16}[a.precision]
    B = Nn.PRECISIONS[a.precision]
    st = Nn.stream_ptr(dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    n, Ho, P, P2, C3 = a.images, 14, 256, 512, 1024
    M = n * Ho * Ho
    g = torch.Generator().manual_seed(2)
    t2 = torch.randn(M, P, generator=g).clamp_min(0).to(bf).to(dev)
    x = torch.randn(n, 2 * Ho, 2 * Ho, P2, generator=g).clamp_min(0).to(bf).to(dev)
    w = (torch.randn(C3, P + P2, generator=g) * (1.0 / (P + P2)) ** 0.5).to(bf).to(dev)
    h3, ones = (torch.randn(C3, generator=g) * 0.1).to(dev), torch.ones(C3, device=dev)
    ws = torch.empty(L.ap_conv_pw_stream_bytes(P + P2, C3), dtype=torch.uint8, device=dev)
    Nn.check(L.ap_conv_pw_pack(B, p(w), P + P2, C3, p(ws), st), "pack")
    wp = torch.empty(L.ap_conv_pair_stream_bytes(P, P2, 0), dtype=torch.uint8, device=dev)
    Nn.check(L.ap_conv_pair_pack(B, p(w), None, P, P2, 0, p(wp), st), "pair pack")
    y, y2 = torch.empty(M, C3, dtype=bf, device=dev), torch.empty(M, C3, dtype=bf, device=dev)

    def timeit(call):
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            call()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / a.iters
    t_pw = timeit(lambda: Nn.check(L.ap_conv_pw_ds_nhwc(B, p(t2), p(x), p(ws), p(ones), p(h3), p(y), n, Ho, P, P2, C3, 2, st), "pw ds"))
    t_pr = timeit(lambda: Nn.check(L.ap_conv_pair_ds_nhwc(B, p(t2), p(x), p(wp), p(ones), p(h3), None, None, p(y2), None, n, Ho, P, P2, 2, 0, st), "pair ds"))
    flops = 2.0 * M * (P + P2) * C3
    print("l3.0.c3d    M=%7d K=%4d N=%4d | conv_pw %7.1f us %6.0f TF/s | pair kernel %7.1f us %6.0f TF/s | x%.2f | equal %s" % (
        M, P + P2, C3, t_pw, flops / t_pw * 1e-6, t_pr, flops / t_pr * 1e-6, t_pr / t_pw, bool(torch.equal(y, y2))))


if __name__ == "__main__":
    main()
    _ap = argparse.ArgumentParser()
    _ap.add_argument("--images", type=int, default=512); _ap.add_argument("--iters", type=int, default=20); _ap.add_argument("--precision", default="f16")
    ds_rows(_ap.parse_args())
    k3_rows(_ap.parse_args())
