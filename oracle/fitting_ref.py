"""ORACLE (test infrastructure) -- PARITY UNPINNED.  CPU restatement (torch + autograd) of the AirPose+ fitting loop,
BASELINE config 5: copenet_real_data/scripts/bundle_adj.py:262-401 (parameters and optimisers :263-290, forward chain
:306-339, losses :341-368, step :398-400; constants :228-247, Geman-McClure :134-136).

The loop's third-party pieces are ABSENT here (human_body_prior @79ee9542, VPoser V02_05 weights, pytorch3d==0.3.0,
torchgeometry==0.1.2: copenet/requirements.txt:8,16,109) and the script needs the real data set, so nothing can be
run for golden vectors; each piece is restated from its published source:
  * VPoser v2 decoder (human_body_prior/models/vposer_model.py): Linear(32,512) - LeakyReLU - Dropout - Linear(512,512)
    - LeakyReLU - Linear(512,126) - ContinousRotReprDecoder (Gram-Schmidt on the 3x2 reshape, stack on the last dim);
    decode() returns pose_body = matrot2aa(rotmats) (rotation_tools.matrot2aa = tgm.rotation_matrix_to_angle_axis of
    the zero-padded 3x4 matrices);
  * BodyModel.forward (human_body_prior/body_model/body_model.py): full_pose = [root_orient | pose_body | jaw, eyes,
    hands = 0], lbs with pose2rot=True (lbs.batch_rodrigues: angle = |r + 1e-8|); Jtr = the 55 posed chain joints;
  * pytorch3d.transforms.rotation_6d_to_matrix: a1 = d6[:3], a2 = d6[3:], Gram-Schmidt, rows = (b1, b2, b3);
  * copenet_real transform_smpl (= copenet utils.py:237-256) and geometry.perspective_projection (:63-91).
Script quirks kept: gmcclure is called with its default sigma = 30 (sigma2d = 40 is never passed, :134,332-345); the
hip confidences are halved IN PLACE on every iteration (:341-342); loss_beta uses the constant zero initial beta
(:357) and so carries no gradient; the first 100 iterations optimise only the rigid poses and beta (:293-295).
"""
import torch
import torch.nn.functional as F

from . import geometry_ref, smplx_ref

W_BETA, W_VPOSER, W_TEMPORAL = 2000.0, 0.05, 1.0          # :243-245
GM_SIGMA = 30.0                                            # default of gmcclure, :134
LR, SWITCH_ITER = 0.01, 100                                # :279-295
NJ = 24                                                    # joints3d[:, :24], :325-334


def vposer_decode(vp, z):
    """VPoser.decode: (L,32) -> pose_body axis-angle (L,21,3) and the decoder's rotation matrices (L,21,3,3)."""
    h = F.leaky_relu(F.linear(z, vp["w1"], vp["b1"]), 0.01)
    h = F.leaky_relu(F.linear(h, vp["w2"], vp["b2"]), 0.01)
    o = F.linear(h, vp["w3"], vp["b3"]).view(-1, 3, 2)
    b1 = F.normalize(o[:, :, 0], dim=1)
    b2 = F.normalize(o[:, :, 1] - (b1 * o[:, :, 1]).sum(1, keepdim=True) * b1, dim=-1)
    R = torch.stack([b1, b2, torch.cross(b1, b2, dim=1)], dim=-1)
    aa = geometry_ref.rotation_matrix_to_angle_axis(F.pad(R, [0, 1]))
    return aa.view(z.shape[0], 21, 3), R.view(z.shape[0], 21, 3, 3)


def lbs_batch_rodrigues(r):
    """human_body_prior / smplx lbs.batch_rodrigues: (N,3) -> (N,3,3), angle = |r + 1e-8|."""
    angle = torch.norm(r + 1e-8, dim=1, keepdim=True)
    d = r / angle
    c, s = torch.cos(angle).unsqueeze(1), torch.sin(angle).unsqueeze(1)
    rx, ry, rz = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    z = torch.zeros_like(rx)
    K = torch.cat([z, -rz, ry, rz, z, -rx, -ry, rx, z], 1).view(-1, 3, 3)
    return torch.eye(3, dtype=r.dtype).unsqueeze(0) + s * K + (1 - c) * torch.bmm(K, K)


def body_joints(model, pose_body, betas):
    """BodyModel.forward(...).Jtr for root_orient = 0, trans = 0, default jaw / eyes / hands: (L,63), (10,) -> (L,55,3)."""
    L = pose_body.shape[0]
    dt = pose_body.dtype
    t = lambda a: torch.as_tensor(a, dtype=dt)
    full = torch.cat([torch.zeros(L, 3, dtype=dt), pose_body, torch.zeros(L, 33 * 3, dtype=dt)], 1)
    R = lbs_batch_rodrigues(full.view(-1, 3)).view(L, 55, 3, 3)
    shapedirs = t(model["shapedirs"])[:, :, :10]
    v_shaped = t(model["v_template"]).unsqueeze(0) + torch.einsum("l,mkl->mk", betas, shapedirs).unsqueeze(0)
    J = smplx_ref.vertices2joints(t(model["J_regressor"]), v_shaped).expand(L, -1, -1)
    posed, _ = smplx_ref.batch_rigid_transform(R, J, model["parents"])
    return posed


def rotation_6d_to_matrix(d6):
    """pytorch3d.transforms.rotation_6d_to_matrix (rows)."""
    a1, a2 = d6[..., :3], d6[..., 3:]
    b1 = F.normalize(a1, dim=-1)
    b2 = F.normalize(a2 - (b1 * a2).sum(-1, keepdim=True) * b1, dim=-1)
    return torch.stack((b1, b2, torch.cross(b1, b2, dim=-1)), dim=-2)


def gmcclure(a, b, sigma=GM_SIGMA):
    x = a - b
    return x ** 2 / (x ** 2 + sigma ** 2)


def loss_terms(vp, model, prm, data, it):
    """One evaluation of the objective at iteration `it` (0-based).  prm: z (L,32), phi0/phi1 (L,6), tau0/tau1 (L,3),
    beta (10,).  data: j2d (2 views, L, 2 detectors, 24, 3 = x, y, conf), robust (L,) bool, intr (2,4 = fx, fy, cx, cy),
    extr (2,3,4)."""
    L = prm["z"].shape[0]
    dt = prm["z"].dtype
    aa, _ = vposer_decode(vp, prm["z"])
    pose_body = aa.reshape(L, 63)
    Jtr = body_joints(model, pose_body, prm["beta"])
    rob = data["robust"]
    hip = torch.ones(NJ, dtype=dt)
    hip[1] = hip[2] = 0.5 ** (it + 1)                       # in-place halving on every iteration, :341-342
    loss_2d = 0.0
    for v in (0, 1):
        Rv = rotation_6d_to_matrix(prm["phi%d" % v])
        tm = torch.cat([Rv, prm["tau%d" % v].unsqueeze(2)], 2)
        _, j3d = geometry_ref.transform_smpl(tm, Jtr, Jtr)
        fx, fy, cx, cy = [data["intr"][v, i] for i in range(4)]
        j2d = geometry_ref.perspective_projection(
            j3d[:, :NJ], data["extr"][v, :, :3].unsqueeze(0).expand(L, -1, -1), data["extr"][v, :, 3].expand(L, -1),
            [fx, fy], torch.stack([cx, cy]))
        for det in (0, 1):
            gt = data["j2d"][v, :, det]
            conf = gt[:, :, 2:] * hip.view(1, NJ, 1)
            loss_2d = loss_2d + (conf[rob] * gmcclure(j2d[rob], gt[rob][:, :, :2])).mean()
    loss_vposer = (prm["z"] * prm["z"]).mean()
    loss_beta = torch.zeros((), dtype=dt)                   # mul(smplxbeta, smplxbeta) of the constant zero init, :357
    rt = rob[:-1] & rob[1:]
    mse = lambda x: ((x[1:] - x[:-1]) ** 2)[rt].mean()
    loss_temporal = 10 * mse(pose_body) + 100 * (mse(prm["phi0"]) + mse(prm["phi1"]) + mse(prm["tau0"]) + mse(prm["tau1"]))
    total = loss_2d + W_BETA * loss_beta + W_VPOSER * loss_vposer + W_TEMPORAL * loss_temporal
    return total, dict(loss_2d=loss_2d, loss_vposer=loss_vposer, loss_temporal=loss_temporal, joints=Jtr)


def loss_and_grads(vp, model, prm, data, it):
    p = {k: v.detach().clone().requires_grad_(True) for k, v in prm.items()}
    total, parts = loss_terms(vp, model, p, data, it)
    total.backward()
    return total.detach(), {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in p.items()}, parts


def fit(vp, model, prm0, data, n_iters=300):
    """The loop of :297-400 with torch.optim.Adam(lr = 0.01); z joins the optimised set at iteration 100."""
    p = {k: v.detach().clone().requires_grad_(True) for k, v in prm0.items()}
    rigid = [p[k] for k in ("phi0", "tau0", "phi1", "tau1", "beta")]
    optim1 = torch.optim.Adam(rigid, lr=LR)
    optim2 = torch.optim.Adam([p["z"]] + rigid, lr=LR)
    losses = []
    for j in range(n_iters):
        optim = optim2 if j >= SWITCH_ITER else optim1
        total, _ = loss_terms(vp, model, p, data, j)
        optim.zero_grad()
        total.backward()
        optim.step()
        losses.append(float(total.detach()))
    return {k: v.detach() for k, v in p.items()}, losses


def synthetic_problem(model, L=64, seed=77, dtype=torch.float64):
    """Seeded stand-in for the gated inputs: random-initialised VPoser decoder (PyTorch Linear init), a ground-truth
    motion, its noisy 2-D observations in both views from two detectors, and a perturbed initial state."""
    g = torch.Generator().manual_seed(seed)
    rnd = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)

    def lin(o, i):
        b = 1.0 / i ** 0.5
        return ((torch.rand(o, i, generator=g, dtype=torch.float64) * 2 - 1) * b), ((torch.rand(o, generator=g, dtype=torch.float64) * 2 - 1) * b)
    vp = {}
    vp["w1"], vp["b1"] = lin(512, 32)
    vp["w2"], vp["b2"] = lin(512, 512)
    vp["w3"], vp["b3"] = lin(126, 512)
    t = torch.linspace(0, 1, L, dtype=torch.float64).unsqueeze(1)
    z_gt = 0.8 * rnd(1, 32) + 0.6 * torch.sin(6.28 * t * torch.rand(1, 32, generator=g, dtype=torch.float64)) * rnd(1, 32)
    gt = {"z": z_gt, "beta": 0.5 * rnd(10)}
    for v in (0, 1):
        gt["phi%d" % v] = torch.tensor([[1.0, 0, 0, 0, -1.0, 0]], dtype=torch.float64) + 0.15 * rnd(1, 6) + 0.05 * t * rnd(1, 6)
        gt["tau%d" % v] = torch.tensor([[0.0, 0.2, 8.0]], dtype=torch.float64) + 0.3 * rnd(1, 3) + 0.4 * t * rnd(1, 3)
    intr = torch.tensor([[1475.0, 1475.0, 960.0, 540.0], [1470.0, 1480.0, 950.0, 545.0]], dtype=torch.float64)
    extr = torch.eye(4, dtype=torch.float64)[:3].unsqueeze(0).repeat(2, 1, 1)           # cam*_extr = identity, :231-235
    data = {"intr": intr, "extr": extr, "robust": torch.rand(L, generator=g) > 0.1}
    with torch.no_grad():
        aa, _ = vposer_decode(vp, gt["z"])
        Jtr = body_joints(model, aa.reshape(L, 63), gt["beta"])
        j2d = torch.zeros(2, L, 2, NJ, 3, dtype=torch.float64)
        for v in (0, 1):
            Rv = rotation_6d_to_matrix(gt["phi%d" % v])
            X = torch.einsum("lij,lkj->lki", Rv, Jtr[:, :NJ]) + gt["tau%d" % v].unsqueeze(1)
            uv = torch.stack([intr[v, 0] * X[..., 0] / X[..., 2] + intr[v, 2], intr[v, 1] * X[..., 1] / X[..., 2] + intr[v, 3]], -1)
            for det in (0, 1):
                j2d[v, :, det, :, :2] = uv + 3.0 * rnd(L, NJ, 2)
                j2d[v, :, det, :, 2] = torch.rand(L, NJ, generator=g, dtype=torch.float64)
    data["j2d"] = j2d
    init = {k: (v + {"z": 0.3, "beta": 0.0, "phi0": 0.05, "phi1": 0.05, "tau0": 0.2, "tau1": 0.2}[k] * rnd(*v.shape)) for k, v in gt.items()}
    init["beta"] = torch.zeros(10, dtype=torch.float64)     # smplxbeta = zeros, :233
    cast = lambda d: {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in d.items()}
    return cast(vp), cast(init), cast(data), cast(gt)
