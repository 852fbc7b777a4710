#!/usr/bin/env python
"""Where do the vertices of a submit() differ from those of __call__ (SMPL-X tail kernel under a concurrent trunk)?  Prints, per
mismatching submit, the differing (body, vertex) pairs: count, vertex % 16, vertex // 16 % 16 (the wave of the group), body % 32."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from airpose_amd import copenet_model, pipeline, smplx, smplx_model, weights as W

def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    prec = sys.argv[2] if len(sys.argv) > 2 else "f16"     # also: fp32, bf16x2, bf16 (stem_direct_kernel / the fp32 kernels beside a second pass)
    dev = torch.device("cuda", 0)
    MEAN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "airpose_amd", "data", "smpl_mean_params.npz")
    net = copenet_model.getcopenet(MEAN, precision=prec).eval()
    net.load_state_dict(W.to_torch(W.copenet_state_dict(1234, MEAN)))
    body = smplx.SMPLX(model_data=smplx_model.make_synthetic_model(4321))
    pipe = pipeline.TwoViewInference(net, body)
    batches = [{k: torch.from_numpy(v).to(dev) for k, v in W.synthetic_inputs(500 + i, B).items()} for i in range(3)]
    want = [{k: v.clone() for k, v in pipe(b).items()} for b in batches]
    torch.cuda.synchronize()
    nbad = 0
    for rep in range(20):
        pend = [pipe.submit(batches[i % 3]) for i in range(6)]
        for i, p in enumerate(pend):
            got = p.synchronize()
            for k in sorted(got):
                if not torch.is_tensor(got[k]) or torch.equal(got[k], want[i % 3][k]):
                    continue
                nbad += 1
                d = got[k] != want[i % 3][k]
                rows = torch.nonzero(d.reshape(d.shape[0], -1).any(1)).flatten().tolist()
                per_c = d.reshape(-1, d.shape[-1]).sum(0).tolist() if d.dim() == 3 else None
                if "j3d" in k:
                    jj = torch.nonzero(d[rows[0]].any(-1)).flatten().tolist()
                    print("   joints", jj, "got", got[k][rows[0], jj[0]].tolist(), "want", want[i % 3][k][rows[0], jj[0]].tolist())
                print(rep, i, k, "n", int(d.sum()), "bodies", rows[:8], "per component", per_c, "max|d| %.3e" % float((got[k] - want[i % 3][k]).abs().max()))
    print("mismatching outputs:", nbad)

main()
