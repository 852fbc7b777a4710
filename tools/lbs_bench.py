#!/usr/bin/env python
"""SMPL-X tail alone (rot6d -> chain -> blend shapes -> skinning -> root transform -> projection; ap_smplx_fwd_fused) at
several body counts, against its HBM line (129 084 algorithmic bytes per body + the model constants once per
launch, peak 8 TB/s).  The bench's own configuration is 512 bodies (2 views x 256 pairs), where the tail is four short
launches; the larger counts show where the kernels themselves sit.   python tools/lbs_bench.py [--bodies 512,4096,16384]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from airpose_amd import smplx, smplx_model  # noqa: E402

BYTES_PER_BODY = 129084
CONST_BYTES = 25.5e6
PEAK = 8.0e12


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bodies", default="512,4096,16384")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--fused", type=int, default=1, help="1: fused contraction + skinning (default); 4: joints stage inside the kernel too; 0: two kernels")
    ap.add_argument("--coherent", type=int, default=0, help="1: synthetic model with vertex ids ordered by their main joint (the real mesh's locality)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    body = smplx.SMPLX(model_data=smplx_model.make_synthetic_model(4321, coherent=bool(args.coherent)))
    body.set_fused(args.fused)
    rows = []
    for n in [int(v) for v in args.bodies.split(",")]:
        g = torch.Generator().manual_seed(n)
        pose = torch.randn(n, 135, generator=g).to(dev)
        betas = (torch.randn(n, 10, generator=g) * 0.5).to(dev)
        cc = torch.tensor([960.0, 540.0]).expand(n, 2).contiguous().to(dev)
        for _ in range(3):
            body.forward_fused(pose, betas, cc)
        body.enable_timing(True)
        body.timing(reset=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            body.forward_fused(pose, betas, cc)
        e1.record()
        torch.cuda.synchronize()
        t = body.timing(reset=True)
        body.enable_timing(False)
        ms = e0.elapsed_time(e1) / args.iters
        k = {s: t[s] / max(t["passes"], 1) for s in ("prep_ms", "blend_gemm_ms", "skin_ms", "joints_ms")}
        alg = n * BYTES_PER_BODY + CONST_BYTES
        skin_bytes = n * 10475 * 3 * 4 * 2                   # skinning alone: reads v_posed, writes vertices
        rows.append({"bodies": n, "ms": ms, "bodies_per_s": n / (ms * 1e-3), "algorithmic_GBps": alg / (ms * 1e-3) / 1e9,
                     "frac_of_hbm_peak": alg / (ms * 1e-3) / PEAK, **k,
                     "skin_kernel_GBps": skin_bytes / (k["skin_ms"] * 1e-3) / 1e9 if k["skin_ms"] else None,
                     "skin_kernel_frac": skin_bytes / (k["skin_ms"] * 1e-3) / PEAK if k["skin_ms"] else None})
    print(json.dumps({"metric": "SMPL-X tail vs HBM line", "unit": "GB/s", "peak_GBps": PEAK / 1e9, "rows": rows}))


if __name__ == "__main__":
    main()
