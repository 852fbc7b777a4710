import sys, time, torch
sys.path.insert(0, '/root/repo')
import os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from airpose_amd import copenet_model, weights as W
dev = torch.device('cuda', 0)
mp = os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'airpose_amd/data/smpl_mean_params.npz')
sd = W.to_torch(W.copenet_state_dict(20240901, mp))
nets = []
for i in range(4):
    n = copenet_model.getcopenet(mp, precision='bf16'); n.load_state_dict(sd); n.eval().to(dev); nets.append(n)
x = torch.randn(512, 3, 224, 224, device=dev)
def single():
    return nets[0].forward_feat_ext(x)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def dual():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1): a = nets[0].forward_feat_ext(x[:256])
    with torch.cuda.stream(s2): b = nets[1].forward_feat_ext(x[256:])
    cur.wait_stream(s1); cur.wait_stream(s2)
    return a, b
ss = [torch.cuda.Stream() for _ in range(4)]
def multi(k):
    def f():
        cur = torch.cuda.current_stream()
        outs = []
        n = 512 // k
        for i in range(k):
            ss[i].wait_stream(cur)
            with torch.cuda.stream(ss[i]): outs.append(nets[i].forward_feat_ext(x[i * n:(i + 1) * n]))
        for i in range(k): cur.wait_stream(ss[i])
        return outs
    return f
def seq():
    return nets[0].forward_feat_ext(x[:256]), nets[0].forward_feat_ext(x[256:])
for name, fn in (('single 512', single), ('two streams 2x256', dual), ('sequential 2x256', seq), ('single 512', single), ('two streams 2x256', dual), ('4 streams x128', multi(4)), ('2 streams (multi)', multi(2)), ('4 streams x128', multi(4))):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(10): fn()
    torch.cuda.synchronize(); print('%-22s %.3f ms' % (name, (time.time() - t0) * 100))
