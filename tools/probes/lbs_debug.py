import sys, torch, numpy as np
sys.path.insert(0, '.')
from airpose_amd import smplx, smplx_model
from oracle import geometry_ref
dev = torch.device('cuda', 0)
md = smplx_model.make_synthetic_model(4321)
body = smplx.SMPLX(model_data=md)
for B in (3, 32, 77, 512):
    gen = torch.Generator().manual_seed(40 + B)
    betas = torch.randn(B, 10, generator=gen)
    R = geometry_ref.rot6d_to_rotmat(torch.randn(B * 22, 6, generator=gen)).reshape(B, 22, 3, 3)
    expr = torch.randn(B, 10, generator=gen) * 0.5; tr = torch.randn(B, 3, generator=gen)
    kw = dict(betas=betas.to(dev), expression=expr.to(dev), body_pose=R[:, 1:].to(dev), global_orient=R[:, :1].to(dev), transl=tr.to(dev), pose2rot=False)
    body.set_fused(0); v2 = body.forward(**kw).vertices.clone()
    body.set_fused(1)
    nbad = 0
    for rep in range(40):
        v1 = body.forward(**kw).vertices
        torch.cuda.synchronize()
        err = (v1 - v2).abs().amax(-1).cpu().numpy()            # [B][V]
        bad = err > 1e-4
        nbad += int(bad.any())
        if bad.any() and nbad <= 2:
            bb, vv = np.nonzero(bad)
            for k in range(min(3, len(bb))):
                print('   b', bb[k], 'v', vv[k], 'got', v1[bb[k], vv[k]].cpu().numpy(), 'want', v2[bb[k], vv[k]].cpu().numpy(), 'tr', tr[bb[k]].numpy())
        if rep == 39: print('B', B, 'runs with errors:', nbad, 'of 40')
        if bad.any() and nbad <= 2: print('B', B, 'rep', rep, 'bad frac %.4f' % bad.mean(), 'bad bodies', np.nonzero(bad.any(1))[0][:10], 'bad verts', np.nonzero(bad.any(0))[0][:12],
              'groups', sorted(set((np.nonzero(bad.any(0))[0] // 16).tolist()))[:10], 'v%16', sorted(set((np.nonzero(bad.any(0))[0] % 16).tolist())))
