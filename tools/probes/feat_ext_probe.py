"""forward_feat_ext on one list of n images: one pass (set_dual_stream(0)) against its two halves as two concurrent passes (default).
Usage: python tools/probes/feat_ext_probe.py [n_images]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from airpose_amd import copenet_model
from airpose_amd import weights as W

MEAN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "airpose_amd", "data", "smpl_mean_params.npz")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda", 0)
net = copenet_model.getcopenet(MEAN, precision="f16").eval()
net.load_state_dict(W.to_torch(W.copenet_state_dict(0, MEAN)))
x = torch.randn(n, 3, 224, 224, device=dev)
res = {}
for dual in (0, 1, 0, 1):
    net.set_dual_stream(dual)
    for _ in range(3):
        f = net.forward_feat_ext(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        f = net.forward_feat_ext(x)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 50
    res.setdefault(dual, f.clone())
    print("dual_stream=%d  %.3f ms per call, %.0f images/s, bit-identical to the first one-pass call: %s"
          % (dual, ms, n / ms * 1e3, torch.equal(f, res[0])))
