#!/usr/bin/env python
"""Is the SMPL-X stage deterministic under a concurrent trunk?  Same (pose, betas) through TwoViewInference._tail on a side stream,
with and without trunk passes on the main stream; every output compared with the first run."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from airpose_amd import copenet_model, pipeline, smplx, smplx_model, weights as W

def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    dev = torch.device("cuda", 0)
    MEAN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "airpose_amd", "data", "smpl_mean_params.npz")
    net = copenet_model.getcopenet(MEAN, precision="f16").eval()
    net.load_state_dict(W.to_torch(W.copenet_state_dict(1234, MEAN)))
    body = smplx.SMPLX(model_data=smplx_model.make_synthetic_model(4321))
    pipe = pipeline.TwoViewInference(net, body)
    batch = {k: torch.from_numpy(v).to(dev) for k, v in W.synthetic_inputs(500, B).items()}
    p0, b0, p1, b1 = pipe.forward_net(batch["im0"], batch["im1"], batch["bb0"], batch["bb1"])
    pose0, betas0 = torch.stack([p0, p1]).clone(), torch.stack([b0, b1]).clone()
    side = torch.cuda.Stream()
    def tail():
        pose, betas = pose0.clone(), betas0.clone()
        return pipe._tail(pose[0], betas[0], pose[1], betas[1], batch, True, False, False)
    ref = {k: v.clone() for k, v in tail().items()}
    torch.cuda.synchronize()
    for mode in ("alone", "under_trunk", "under_copy"):
        bad = {}
        noise = torch.empty(64 << 20, dtype=torch.float32, device=dev)
        for rep in range(60):
            if mode == "under_trunk":
                net.forward_feat_ext_twoview(batch["im0"], batch["im1"])
            elif mode == "under_copy":
                noise.add_(1.0)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                if mode != "alone":
                    torch.cuda._sleep(200000 + 150000 * (rep % 7))
                got = tail()
            side.synchronize()
            torch.cuda.synchronize()
            for k, v in ref.items():
                if not torch.equal(got[k], v):
                    d = (got[k] != v)
                    rows = torch.nonzero(d.reshape(d.shape[0], -1).any(1)).flatten().tolist()
                    bad.setdefault(k, []).append((rep, int(d.sum()), rows[:6], float((got[k] - v).abs().max())))
        print(mode, {k: (len(v), v[:3]) for k, v in bad.items()})

main()
