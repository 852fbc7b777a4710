#!/usr/bin/env python
"""Co-run probe: what two trunk passes on two streams cost when they run the SAME kernel together (today's lock-step: pair || pair,
then slab || slab) against COMPLEMENTARY kernels together (pair || slab, then slab || pair: an HBM-bound beside an MFMA-bound one).
256 images per pass, layer2 and layer3 shapes.   python tools/probes/corun_probe.py"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from airpose_amd import _native as N
dev = torch.device("cuda", 0)
L = N.lib()
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
BF = N.PRECISIONS["bf16"]
bf = torch.bfloat16

def make(H, P, N1, n=256):
    M, C3 = n * H * H, 4 * P
    d = {}
    d["t2"] = torch.randn(M, P, device=dev).clamp_min(0).to(bf); d["x"] = torch.randn(M, C3, device=dev).clamp_min(0).to(bf)
    w3 = (torch.randn(C3, P, device=dev) * (2.0 / P) ** 0.5).to(bf); w1 = (torch.randn(N1, C3, device=dev) * (2.0 / C3) ** 0.5).to(bf)
    d["s3"], d["h3"] = torch.rand(C3, device=dev) + 0.5, torch.randn(C3, device=dev) * 0.1
    d["s1"], d["h1"] = torch.rand(N1, device=dev) + 0.5, torch.randn(N1, device=dev) * 0.1
    d["out"], d["t1n"] = torch.empty(M, C3, dtype=bf, device=dev), torch.empty(M, N1, dtype=bf, device=dev)
    d["ws"] = torch.empty(L.ap_conv_pair_stream_bytes(P, 0, N1), dtype=torch.uint8, device=dev)
    N.check(L.ap_conv_pair_pack(BF, p(w3), p(w1), P, 0, N1, p(d["ws"]), N.stream_ptr(dev)), "pack")
    # the 3x3 of the block
    d["w2"] = (torch.randn(128 * ((P + 127) // 128), 3, 3, P, device=dev) * (2.0 / (9 * P)) ** 0.5).to(bf)
    d["s2"], d["h2"] = torch.rand(d["w2"].shape[0], device=dev) + 0.5, torch.randn(d["w2"].shape[0], device=dev) * 0.1
    d["t1"] = torch.randn(n, H, H, P, device=dev).clamp_min(0).to(bf); d["t2o"] = torch.empty(n, H, H, P, dtype=bf, device=dev)
    d["meta"] = (n, H, P, N1, M)
    return d

def pair(d, st):
    n, H, P, N1, M = d["meta"]
    N.check(L.ap_conv_pair_nhwc(BF, p(d["t2"]), p(d["ws"]), p(d["s3"]), p(d["h3"]), p(d["x"]), p(d["s1"]), p(d["h1"]), p(d["out"]), p(d["t1n"]), M, P, N1,
                                ctypes.c_void_p(st.cuda_stream)), "pair")

def slab(d, st):
    n, H, P, N1, M = d["meta"]
    N.check(L.ap_conv2d_nhwc(BF, p(d["t1"]), p(d["w2"]), p(d["s2"]), p(d["h2"]), None, p(d["t2o"]), n, H, H, P, P, 3, 1, 1, 1, ctypes.c_void_p(st.cuda_stream)), "slab")

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters

sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
cur = torch.cuda.current_stream()
for tag, H, P, N1 in (("layer2", 28, 128, 128), ("layer3", 14, 256, 256)):
    a, b = make(H, P, N1), make(H, P, N1)
    def both(fa, fb):
        def run():
            sA.wait_stream(cur); sB.wait_stream(cur)
            fa(a, sA); fb(b, sB)
            cur.wait_stream(sA); cur.wait_stream(sB)
        return run
    t_p = timeit(lambda: pair(a, cur)); t_s = timeit(lambda: slab(a, cur))
    t_pp = timeit(both(pair, pair)); t_ss = timeit(both(slab, slab)); t_ps = timeit(both(pair, slab))
    print("%s (256 images per pass): alone pair %.1f slab %.1f us | pair||pair %.1f + slab||slab %.1f = %.1f us per block of both passes | "
          "pair||slab %.1f x 2 = %.1f us" % (tag, t_p, t_s, t_pp, t_ss, t_pp + t_ss, t_ps, 2 * t_ps))
