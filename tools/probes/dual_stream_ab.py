import sys, os, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from airpose_amd import copenet_model, weights as W
dev = torch.device('cuda', 0)
mp = os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'airpose_amd/data/smpl_mean_params.npz')
sd = W.to_torch(W.copenet_state_dict(20240901, mp))
net = copenet_model.getcopenet(mp, precision='bf16'); net.load_state_dict(sd); net.eval()
b = {k: torch.from_numpy(v).to(dev) for k, v in W.synthetic_inputs(1234, 256).items()}
pos = torch.tensor([0., 0., .5], device=dev).expand(256, 3).contiguous()
def fwd(): return net(b['im0'], b['im1'], b['bb0'], b['bb1'], pos, pos, iters=3)
outs = {}
for rep in range(3):
    for on in [int(v) for v in os.environ.get('MODES', '1,0').split(',')]:
        fwd(); net.set_dual_stream(on)
        for _ in range(3): o = fwd()
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(20): o = fwd()
        torch.cuda.synchronize(); print('dual=%d  %.3f ms/step' % (on, (time.time() - t0) * 50))
        outs[on] = [t.clone() for t in o]
ks = sorted(outs); print('bit-identical:', all(torch.equal(a, c) for k in ks[1:] for a, c in zip(outs[ks[0]], outs[k])))
