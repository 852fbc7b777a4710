#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REAL reference modules (build container only).

The Python reference cannot travel to the GPU box, so this script imports it from
/root/reference here, feeds it the weights/inputs of OUR seeded generators
(airpose_amd.weights) and stores only data: small inputs, seeds, checksums, sampled
activations and the reference's outputs.  No reference source is copied.

  python tools/make_golden.py            # rewrites tests/golden/{copenet_b2,hmr_b1,geometry}.npz
"""
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SRC = "/root/reference/copenet/src"
REF_SRC_REAL = "/root/reference/copenet_real/src"
sys.path.insert(0, REPO)
OUT = os.path.join(REPO, "tests", "golden")
MEAN = os.path.join(REPO, "airpose_amd", "data", "smpl_mean_params.npz")
WSEED, ISEED = 20240901, 1234


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference():
    # torchvision is only used for ImageNet init (model_copenet.py:236-238); cv2/torchgeometry are
    # module-level imports of utils.py that transform_smpl never touches.
    tv = _stub("torchvision")
    tvm = _stub("torchvision.models")
    tvr = _stub("torchvision.models.resnet")
    tv.models, tvm.resnet = tvm, tvr
    _stub("cv2")
    _stub("torchgeometry")
    _stub("pytorch_lightning", LightningModule=torch.nn.Module)     # model_copenet_sep.py subclasses pl.LightningModule
    sys.path.insert(0, REF_SRC)
    sys.path.insert(0, REF_SRC_REAL)
    from copenet.models import model_copenet, model_hmr
    from copenet.utils import geometry
    from copenet.utils import utils as ref_utils
    return model_copenet, model_hmr, geometry, ref_utils


def sample(t, n=512):
    f = t.detach().reshape(-1)
    idx = torch.linspace(0, f.numel() - 1, n).long()
    return f[idx].numpy().copy(), idx.numpy().copy()


def main():
    from airpose_amd import weights as W
    torch.set_num_threads(8)
    torch.manual_seed(0)
    model_copenet, model_hmr, geometry, ref_utils = import_reference()
    os.makedirs(OUT, exist_ok=True)

    # ------------------------------------------------------------------ copenet two-view, B=2
    sd = W.to_torch(W.copenet_state_dict(WSEED, MEAN))
    net = model_copenet.copenet(model_copenet.Bottleneck, [3, 4, 6, 3], MEAN).eval()
    missing = net.load_state_dict(sd, strict=True)
    B = 2
    inp = {k: torch.from_numpy(v) for k, v in W.synthetic_inputs(ISEED, B).items()}
    pos = torch.tensor([0.0, 0.0, 10.0]).expand(B, -1).clone() * 0.05
    g = {"weights_seed": WSEED, "inputs_seed": ISEED, "batch": B,
         "state_dict_keys": np.array(list(net.state_dict().keys())),
         "im0_sum": inp["im0"].double().sum().item(), "im1_sum": inp["im1"].double().sum().item(),
         "bb0": inp["bb0"].numpy(), "bb1": inp["bb1"].numpy(), "init_position": pos.numpy()}
    g["im0_sample"], g["im_sample_idx"] = sample(inp["im0"])
    acts = {}
    hooks = [net.maxpool.register_forward_hook(lambda m, i, o: acts.__setitem__("stem", o))]
    for li in (1, 2, 3, 4):
        hooks.append(getattr(net, "layer%d" % li).register_forward_hook(
            lambda m, i, o, li=li: acts.__setitem__("layer%d" % li, o)))
    with torch.no_grad():
        xf0 = net.forward_feat_ext(inp["im0"])
        for k, v in acts.items():
            g["act0_%s" % k], g["act_idx_%s" % k] = sample(v)
            g["act0_%s_absmean" % k] = v.abs().double().mean().item()
        xf1 = net.forward_feat_ext(inp["im1"])
        for h in hooks:
            h.remove()
        g["xf0"], g["xf1"] = xf0.numpy(), xf1.numpy()
        for it in (1, 2, 3):
            p0, b0, p1, b1 = net(inp["im0"], inp["im1"], inp["bb0"], inp["bb1"], pos, pos, iters=it)
            g["pose0_it%d" % it], g["betas0_it%d" % it] = p0.numpy(), b0.numpy()
            g["pose1_it%d" % it], g["betas1_it%d" % it] = p1.numpy(), b1.numpy()
        # caller-supplied initial state (model_copenet.py:121-136)
        th0 = torch.randn(1, 144) * 0.3
        th1 = torch.randn(1, 144) * 0.3
        s0, s1 = torch.randn(B, 10) * 0.5, torch.randn(B, 10) * 0.5
        p0, b0, p1, b1 = net(inp["im0"], inp["im1"], inp["bb0"], inp["bb1"], pos, pos,
                             init_theta0=th0, init_theta1=th1, init_shape0=s0, init_shape1=s1, iters=2)
        g.update(ci_theta0=th0.numpy(), ci_theta1=th1.numpy(), ci_shape0=s0.numpy(), ci_shape1=s1.numpy(),
                 ci_pose0=p0.numpy(), ci_betas0=b0.numpy(), ci_pose1=p1.numpy(), ci_betas1=b1.numpy())
    np.savez_compressed(os.path.join(OUT, "copenet_b2.npz"), **g)
    print("copenet_b2: xf0 absmean %.4f pose0 %s" % (np.abs(g["xf0"]).mean(), g["pose0_it3"][0, :9]))

    # ------------------------------------------------------------------ hmr single view, B=1 (Config 1)
    sdh = W.to_torch(W.copenet_state_dict(WSEED + 1, MEAN, variant="hmr"))
    hnet = model_hmr.copenet(model_hmr.Bottleneck, [3, 4, 6, 3], MEAN).eval()
    hnet.load_state_dict(sdh, strict=True)
    x = torch.from_numpy(W.synthetic_inputs(ISEED + 1, 1)["im0"])
    with torch.no_grad():
        rotmat, betas, cam = hnet(x, iters=3)
    np.savez_compressed(os.path.join(OUT, "hmr_b1.npz"), weights_seed=WSEED + 1, inputs_seed=ISEED + 1,
                        im_sum=x.double().sum().item(), rotmat=rotmat.numpy(), betas=betas.numpy(), cam=cam.numpy())
    print("hmr_b1: betas", betas.numpy()[0, :4])

    # ------------------------------------------------------------------ copenet_singleview baseline, B=1
    from copenet.models import model_copenet_singleview
    sds = W.to_torch(W.copenet_state_dict(WSEED + 4, MEAN, variant="singleview"))
    snet = model_copenet_singleview.copenet(model_copenet_singleview.Bottleneck, [3, 4, 6, 3], MEAN).eval()
    snet.load_state_dict(sds, strict=True)
    si = W.synthetic_inputs(ISEED + 4, 1)
    with torch.no_grad():
        sp, sb = snet(torch.from_numpy(si["im0"]), torch.from_numpy(si["bb0"]), pos[:1], iters=3)
    np.savez_compressed(os.path.join(OUT, "singleview_b1.npz"), weights_seed=WSEED + 4, inputs_seed=ISEED + 4,
                        state_dict_keys=np.array(list(snet.state_dict().keys())), pose=sp.numpy(), betas=sb.numpy(),
                        init_position=pos[:1].numpy())
    print("singleview_b1: pose", sp.numpy()[0, :6])

    # ------------------------------------------------------------------ muhmr two-view baseline, B=1
    from copenet.models import model_muhmr
    sdm = W.to_torch(W.copenet_state_dict(WSEED + 5, MEAN, variant="muhmr"))
    mnet = model_muhmr.copenet(model_muhmr.Bottleneck, [3, 4, 6, 3], MEAN).eval()
    mnet.load_state_dict(sdm, strict=True)
    mi = W.synthetic_inputs(ISEED + 5, 1)
    with torch.no_grad():
        mo = mnet(torch.from_numpy(mi["im0"]), torch.from_numpy(mi["im1"]), iters=3)
    np.savez_compressed(os.path.join(OUT, "muhmr_b1.npz"), weights_seed=WSEED + 5, inputs_seed=ISEED + 5,
                        state_dict_keys=np.array(list(mnet.state_dict().keys())),
                        **{k: v.numpy() for k, v in zip(("pose0", "betas0", "cam0", "pose1", "betas1", "cam1"), mo)})
    print("muhmr_b1: cam0", mo[2].numpy()[0])

    # ------------------------------------------------------------------ copenet_sep (two weight sets; IEF from features)
    # The two trunks are ordinary ResNet-50s (pinned above); the fixture pins the sep driver + its asymmetric
    # forward_reg by feeding the golden trunk features through the reference model with forward_feat_ext patched out.
    from copenet_real.models import model_copenet_sep
    sep = model_copenet_sep.copenet_sep(model_copenet_sep.Bottleneck, [3, 4, 6, 3], MEAN).eval()
    sep.copenet0.load_state_dict(W.to_torch(W.copenet_state_dict(WSEED + 2, MEAN)), strict=True)
    sep.copenet1.load_state_dict(W.to_torch(W.copenet_state_dict(WSEED + 3, MEAN)), strict=True)
    sep.copenet0.forward_feat_ext = lambda x: torch.from_numpy(g["xf0"])
    sep.copenet1.forward_feat_ext = lambda x: torch.from_numpy(g["xf1"])
    gs = {"weights_seed0": WSEED + 2, "weights_seed1": WSEED + 3,
          "state_dict_keys": np.array(list(sep.state_dict().keys()))}
    with torch.no_grad():
        for it in (1, 3):
            p0, b0, p1, b1 = sep(inp["im0"], inp["im1"], inp["bb0"], inp["bb1"], pos, pos, iters=it)
            gs["pose0_it%d" % it], gs["betas0_it%d" % it] = p0.numpy(), b0.numpy()
            gs["pose1_it%d" % it], gs["betas1_it%d" % it] = p1.numpy(), b1.numpy()
    np.savez_compressed(os.path.join(OUT, "copenet_sep_b2.npz"), **gs)
    print("copenet_sep_b2: pose1", gs["pose1_it3"][0, :6])

    # ------------------------------------------------------------------ geometry helpers
    rs = np.random.RandomState(7)
    x6 = torch.from_numpy(rs.standard_normal((5, 132)).astype(np.float32))
    x6[0, :6] = torch.tensor([1.0, 0, 0, 1, 0, 0])            # -> identity
    x6[1, :6] = 0.0                                           # degenerate: eps path of F.normalize
    pts = torch.from_numpy((rs.standard_normal((3, 127, 3)) + np.array([0, 0, 8.0])).astype(np.float32))
    cc = torch.from_numpy(rs.uniform(400, 1000, (3, 2)).astype(np.float32))
    Rm = geometry.rot6d_to_rotmat(torch.from_numpy(rs.standard_normal((3, 6)).astype(np.float32)))
    tt = torch.from_numpy(rs.standard_normal((3, 3)).astype(np.float32))
    verts = torch.from_numpy(rs.standard_normal((3, 200, 3)).astype(np.float32))
    aa = torch.from_numpy(rs.standard_normal((6, 3)).astype(np.float32))
    with torch.no_grad():
        g2 = dict(
            rot6d_in=x6.numpy(), rot6d_out=geometry.rot6d_to_rotmat(x6).numpy(),
            proj_points=pts.numpy(), proj_center=cc.numpy(), proj_focal=np.array([1475.0, 1475.0]),
            proj_out=geometry.perspective_projection(
                pts, torch.eye(3).expand(3, 3, 3), torch.zeros(3, 3), [1475, 1475], cc.unsqueeze(0)).numpy(),
            proj_rt_R=Rm.numpy(), proj_rt_t=tt.numpy(),
            proj_rt_out=geometry.perspective_projection(pts, Rm, tt + torch.tensor([0, 0, 5.0]),
                                                        [1000.0, 1100.0], cc).numpy(),
            tf_mat=torch.cat([Rm, tt.unsqueeze(2)], 2).numpy(), tf_verts=verts.numpy(),
            tf_joints=pts.numpy(), rodrigues_in=aa.numpy(), rodrigues_out=geometry.batch_rodrigues(aa).numpy())
        v, j, _, _ = ref_utils.transform_smpl(torch.cat([Rm, tt.unsqueeze(2)], 2), verts, pts)
        g2["tf_verts_out"], g2["tf_joints_out"] = v.numpy(), j.numpy()
    np.savez_compressed(os.path.join(OUT, "geometry.npz"), **g2)
    print("geometry: ok")


if __name__ == "__main__":
    main()
