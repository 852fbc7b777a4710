#!/usr/bin/env python
"""BASELINE config 5: AirPose+ fitting loop, batch (sequence length) 64, end-to-end latency on one MI355X.

  python tools/fit_bench.py [--frames 64] [--iters 300] [--cpu-iters 20]

Prints one JSON line: latency of the full 300-iteration fit (ap_fit_run, inputs resident on the device), per-iteration
time, and the same loop of the CPU oracle (torch autograd, host cores) on a bounded number of iterations."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from airpose_amd import smplx as smplx_mod, smplx_model  # noqa: E402
from airpose_amd.fitting import AirPosePlusFitter  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--cpu-iters", type=int, default=20)
    args = ap.parse_args()
    from oracle import fitting_ref          # synthetic problem generator + CPU baseline (bench leg only)
    dev = torch.device("cuda", 0)
    md = smplx_model.make_synthetic_model(4321)
    body = smplx_mod.SMPLX(model_data=md, batch_size=args.frames, create_transl=False).to(dev)
    vp, init, data, _ = fitting_ref.synthetic_problem(md, L=args.frames, seed=77, dtype=torch.float64)
    fitter = AirPosePlusFitter(vp, body, dev)
    dd = {k: (v.to(dev).float() if v.is_floating_point() else v) for k, v in data.items()}
    st = {k: v.to(dev).float() for k, v in init.items()}
    run = lambda n: fitter.run(st, dd["j2d"], dd["robust"], dd["intr"], dd["extr"][:, :3], n_iters=n, want_loss=True)
    run(5)
    torch.cuda.synchronize()
    t0 = time.time()
    out, hist = run(args.iters)
    torch.cuda.synchronize()
    ms = (time.time() - t0) * 1e3
    cpu_ms = None
    if args.cpu_iters:
        cores = min(16, os.cpu_count() or 1)         # tiny tensors: more threads only add contention (256 threads: 25 s / iteration)
        torch.set_num_threads(cores)
        f32 = lambda d: {k: (v.float() if v.is_floating_point() else v) for k, v in d.items()}
        t0 = time.time()
        fitting_ref.fit(f32(vp), md, f32(init), f32(data), n_iters=args.cpu_iters)
        cpu_ms = (time.time() - t0) * 1e3 / args.cpu_iters
    print(json.dumps({
        "metric": "AirPose+ fitting loop end-to-end latency (BASELINE config 5)", "value": ms, "unit": "ms",
        "higher_is_better": False, "n_gpus": 1, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "bundle_adj.py:262-401: %d Adam steps, sequence of %d frames, 2 views x 2 detectors x 24 joints"
                               % (args.iters, args.frames)},
        "ms_per_iteration": ms / args.iters, "loss_first": float(hist[0, :3].sum()), "loss_last": float(hist[-1, :3].sum()),
        "cpu_baseline": None if cpu_ms is None else {"value": cpu_ms * args.iters, "unit": "ms", "cores": cores, "kind": "port",
                                                     "sample": "%d iterations of the torch-autograd oracle, extrapolated to %d" % (args.cpu_iters, args.iters)}}))


if __name__ == "__main__":
    main()
