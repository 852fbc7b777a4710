#!/usr/bin/env python
"""Two batches in flight: does the latency-bound end of a step (avg-pool, regressor, SMPL-X tail: ~0.19 ms of 6.1) hide under the
next step's stem / layer1 when consecutive steps are issued on alternating streams (two handles, two workspaces)?
  python tools/probes/inflight_probe.py [--batch 256] [--steps 40] [--reps 3]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from airpose_amd import copenet_model, pipeline, smplx, smplx_model, weights as W  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--precision", default="f16")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    MEAN = os.path.join(os.path.dirname(copenet_model.__file__), "data", "smpl_mean_params.npz")
    sd = W.to_torch(W.copenet_state_dict(20240901, MEAN))
    md = smplx_model.make_synthetic_model(4321)
    pipes = []
    for _ in range(2):
        net = copenet_model.getcopenet(MEAN, precision=args.precision).eval()
        net.load_state_dict(sd)
        pipes.append(pipeline.TwoViewInference(net, smplx.SMPLX(model_data=md), iters=3))
    batch = {k: torch.from_numpy(v).to(dev) for k, v in W.synthetic_inputs(1234, args.batch).items()}
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]

    def run(inflight, steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            k = i % inflight
            with torch.cuda.stream(streams[k]):
                pipes[k](batch, want_rotmat=True)
        torch.cuda.synchronize()
        return args.batch * steps / (time.perf_counter() - t0)

    for inflight in (1, 2):
        run(inflight, 6)
    for rep in range(args.reps):
        for inflight in (1, 2):
            print("in flight %d  r%d: %.0f pairs/s" % (inflight, rep, run(inflight, args.steps)), flush=True)


if __name__ == "__main__":
    main()
