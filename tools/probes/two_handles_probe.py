"""Two network handles alive in one process (the runtime deals its hardware queues to all streams of the process): serving-loop rate
of the SECOND handle at B pairs, network only.  Usage: python tools/probes/two_handles_probe.py [B]"""
import os, sys, time
import torch
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, R)
from airpose_amd import copenet_model, pipeline
from airpose_amd import weights as W

MEAN = os.path.join(R, "airpose_amd", "data", "smpl_mean_params.npz")
dev = torch.device("cuda", 0)
sd = W.to_torch(W.copenet_state_dict(0, MEAN))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
b = {k: torch.from_numpy(v).to(dev) for k, v in W.synthetic_inputs(99, B).items()}
keep = []
for i, prec in enumerate(("f16", "bf16", "f16")):
    net = copenet_model.getcopenet(MEAN, precision=prec).eval()
    net.load_state_dict(sd)
    keep.append(net)                                         # earlier handles (and their streams) stay alive
    pipe = pipeline.TwoViewInference(net, None, iters=3)
    for form in ("call", "submit"):
        f = (lambda: pipe.forward_net(b["im0"], b["im1"], b["bb0"], b["bb1"])) if form == "call" else \
            (lambda: pipe.submit_net(b["im0"], b["im1"], b["bb0"], b["bb1"]))
        for _ in range(4):
            o = f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(60):
            o = f()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("handle %d (%s) %s: %.3f ms/step %.0f pairs/s" % (i, prec, form, dt / 60 * 1e3, B * 60 / dt))
