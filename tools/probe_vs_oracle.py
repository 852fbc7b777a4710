#!/usr/bin/env python3
"""How good a predictor is the GPU parity probe (ap_net_parity_probe: the handle's trunk against an exact-fp32 trunk of the same
weights, network outputs only, no CPU involved) of what the fp32 CPU oracle says about the WHOLE pipeline?  For a grid of synthetic
checkpoints (weight seeds x BatchNorm-statistics recipes) and the two 16-bit storage types: probe error | worst slice-max error
against the oracle on 16 fresh pairs | do the two agree on the 1e-4 bar.   python tools/probe_vs_oracle.py [--seeds 1,2,3,4]
(test infrastructure: imports oracle/)"""
import argparse
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from airpose_amd import _native as Nn  # noqa: E402
from airpose_amd import copenet_model, pipeline, smplx, smplx_model  # noqa: E402
from airpose_amd import weights as W  # noqa: E402
from oracle import pipeline_ref  # noqa: E402

MEAN = os.path.join(REPO, "airpose_amd", "data", "smpl_mean_params.npz")
KEYS = ("pred_pose0", "pred_pose1", "pred_betas0", "pred_betas1", "pred_j3d_cam0", "pred_j3d_cam1", "pred_j2d_cam0", "pred_j2d_cam1",
        "pred_vertices_cam0", "pred_vertices_cam1")


def worst(got, want):
    w = 0.0
    for k in KEYS:
        a, b = got[k].double().cpu().numpy(), want[k].double().numpy()
        parts = [(a[:, :3], b[:, :3]), (a[:, 3:], b[:, 3:])] if "pose" in k else [(a, b)]
        for x, y in parts:
            w = max(w, float(np.abs(x - y).max() / max(np.abs(y).max(), 1e-30)))
    return w


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="1,2,3,4")
    ap.add_argument("--pairs", type=int, default=16)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    md = smplx_model.make_synthetic_model(4321)
    body = smplx.SMPLX(model_data=md)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    rows, agree, tot = [], 0, 0
    nets = {m: copenet_model.getcopenet(MEAN, precision=m).eval() for m in ("f16", "bf16", "bf16x2")}
    for seed in [int(v) for v in a.seeds.split(",")]:
        for bn in ("default", "wide", "survey"):
            sd = W.to_torch(W.copenet_state_dict(seed, MEAN, bn=bn))
            inp = {k: torch.from_numpy(v) for k, v in W.synthetic_inputs(100 + seed, a.pairs).items()}
            with torch.no_grad():
                want = pipeline_ref.infer(sd, md, inp["im0"], inp["im1"], inp["bb0"], inp["bb1"], inp["intr0"], inp["intr1"])
            gin = {k: v.to(dev) for k, v in inp.items()}
            for m, net in nets.items():
                net.load_state_dict(sd)
                try:
                    pr = net.parity_probe(8)["max_rel_err"]
                    got = pipeline.TwoViewInference(net, body)(gin)
                    orc = worst(got, want)
                    if m == "f16":
                        net.range_status()
                except (Nn.RangeError, RuntimeError) as e:
                    if "fp16 range" not in str(e):
                        raise
                    pr = orc = float("inf")
                    try:
                        net.range_status(reset=True)
                    except Nn.RangeError:
                        pass
                ok = (pr < 1e-4) == (orc < 1e-4)
                agree += ok
                tot += 1
                rows.append((seed, bn, m, pr, orc, ok))
                print("seed %d  bn %-7s  %-6s  probe %.2e  oracle %.2e  %s" % (seed, bn, m, pr, orc, "agree" if ok else "DISAGREE"), flush=True)
    print("probe and oracle agree on the 1e-4 bar in %d of %d (mode, checkpoint) cells" % (agree, tot))
    r = np.array([[p, o] for (_, _, _, p, o, _) in rows if np.isfinite(p) and np.isfinite(o) and p > 0])
    print("oracle / probe error ratio over the finite cells: median %.2f, min %.2f, max %.2f" % (np.median(r[:, 1] / r[:, 0]), (r[:, 1] / r[:, 0]).min(), (r[:, 1] / r[:, 0]).max()))


if __name__ == "__main__":
    main()
