"""Pin the oracle to the reference: every function in oracle/ vs tests/golden (made by
tools/make_golden.py from the imported reference modules).  CPU only."""
import numpy as np
import torch

from conftest import MEAN_PARAMS, pose_rel_errs, rel_err
from oracle import copenet_ref, geometry_ref

TOL = 2e-6   # same torch kernels on the same CPU arithmetic; only op fusion order may differ


def pose_err(a, b):
    return max(pose_rel_errs(a, b).values())


def test_generators_reproduce_golden_inputs(golden, copenet_inputs):
    g = golden["copenet_b2"]
    assert abs(copenet_inputs["im0"].double().sum().item() - float(g["im0_sum"])) < 1e-6
    assert abs(copenet_inputs["im1"].double().sum().item() - float(g["im1_sum"])) < 1e-6
    flat = copenet_inputs["im0"].reshape(-1)[torch.from_numpy(g["im_sample_idx"])]
    assert np.array_equal(flat.numpy(), g["im0_sample"])
    assert np.array_equal(copenet_inputs["bb0"].numpy(), g["bb0"])


def test_state_dict_keys_match_reference(golden, copenet_sd):
    ref_keys = [str(k) for k in golden["copenet_b2"]["state_dict_keys"]]
    assert sorted(copenet_sd.keys()) == sorted(ref_keys)
    assert len(ref_keys) == 331


def test_trunk_matches_reference(golden, copenet_sd, copenet_inputs):
    g = golden["copenet_b2"]
    with torch.no_grad():
        taps = {}
        xf0 = copenet_ref.forward_feat_ext(copenet_inputs["im0"], copenet_sd, taps)
        xf1 = copenet_ref.forward_feat_ext(copenet_inputs["im1"], copenet_sd)
    assert rel_err(xf0.numpy(), g["xf0"]) < TOL
    assert rel_err(xf1.numpy(), g["xf1"]) < TOL
    for k, v in taps.items():
        s = v.reshape(-1)[torch.from_numpy(g["act_idx_%s" % k])].numpy()
        assert rel_err(s, g["act0_%s" % k]) < TOL, k


def test_ief_matches_reference(golden, copenet_sd, copenet_inputs):
    g = golden["copenet_b2"]
    xf0, xf1 = torch.from_numpy(g["xf0"]), torch.from_numpy(g["xf1"])
    pos = torch.from_numpy(g["init_position"])
    with torch.no_grad():
        for it in (1, 2, 3):
            p0, b0, p1, b1 = copenet_ref.ief(copenet_sd, xf0, xf1, copenet_inputs["bb0"], copenet_inputs["bb1"],
                                             pos, pos, iters=it)
            assert pose_err(p0.numpy(), g["pose0_it%d" % it]) < TOL
            assert pose_err(p1.numpy(), g["pose1_it%d" % it]) < TOL
            assert rel_err(b0.numpy(), g["betas0_it%d" % it]) < TOL
            assert rel_err(b1.numpy(), g["betas1_it%d" % it]) < TOL
        p0, b0, p1, b1 = copenet_ref.ief(
            copenet_sd, xf0, xf1, copenet_inputs["bb0"], copenet_inputs["bb1"], pos, pos,
            init_theta0=torch.from_numpy(g["ci_theta0"]), init_theta1=torch.from_numpy(g["ci_theta1"]),
            init_shape0=torch.from_numpy(g["ci_shape0"]), init_shape1=torch.from_numpy(g["ci_shape1"]), iters=2)
    assert pose_err(p0.numpy(), g["ci_pose0"]) < TOL and pose_err(p1.numpy(), g["ci_pose1"]) < TOL
    assert rel_err(b0.numpy(), g["ci_betas0"]) < TOL and rel_err(b1.numpy(), g["ci_betas1"]) < TOL


def test_full_forward_matches_reference(golden, copenet_sd, copenet_inputs):
    g = golden["copenet_b2"]
    pos = torch.from_numpy(g["init_position"])
    with torch.no_grad():
        p0, b0, p1, b1 = copenet_ref.copenet_forward(copenet_sd, copenet_inputs["im0"], copenet_inputs["im1"],
                                                     copenet_inputs["bb0"], copenet_inputs["bb1"], pos, pos, iters=3)
    assert pose_err(p0.numpy(), g["pose0_it3"]) < TOL and rel_err(b1.numpy(), g["betas1_it3"]) < TOL


def test_hmr_config1_matches_reference(golden):
    """BASELINE config 0: hmr single view, batch 1, CPU."""
    from airpose_amd import weights as W
    g = golden["hmr_b1"]
    sd = W.to_torch(W.copenet_state_dict(int(g["weights_seed"]), MEAN_PARAMS, variant="hmr"))
    x = torch.from_numpy(W.synthetic_inputs(int(g["inputs_seed"]), 1)["im0"])
    assert abs(x.double().sum().item() - float(g["im_sum"])) < 1e-6
    with torch.no_grad():
        rotmat, betas, cam = copenet_ref.hmr_forward(sd, x, iters=3)
    assert rel_err(rotmat.numpy(), g["rotmat"]) < 1e-5
    assert rel_err(betas.numpy(), g["betas"]) < TOL and rel_err(cam.numpy(), g["cam"]) < TOL


def test_geometry_matches_reference(golden):
    g = golden["geometry"]
    R = geometry_ref.rot6d_to_rotmat(torch.from_numpy(g["rot6d_in"]))
    assert np.allclose(R.numpy(), g["rot6d_out"], rtol=0, atol=1e-6)
    assert np.allclose(R[0].numpy(), np.eye(3), atol=1e-7)            # [1,0,0,1,0,0] -> I
    pts = torch.from_numpy(g["proj_points"])
    out = geometry_ref.perspective_projection(pts, torch.eye(3).expand(3, 3, 3), torch.zeros(3, 3),
                                              [1475, 1475], torch.from_numpy(g["proj_center"]).unsqueeze(0))
    assert rel_err(out.numpy(), g["proj_out"]) < TOL
    out = geometry_ref.perspective_projection(pts, torch.from_numpy(g["proj_rt_R"]),
                                              torch.from_numpy(g["proj_rt_t"]) + torch.tensor([0, 0, 5.0]),
                                              [1000.0, 1100.0], torch.from_numpy(g["proj_center"]))
    assert rel_err(out.numpy(), g["proj_rt_out"]) < TOL
    v, j = geometry_ref.transform_smpl(torch.from_numpy(g["tf_mat"]), torch.from_numpy(g["tf_verts"]),
                                       torch.from_numpy(g["tf_joints"]))
    assert rel_err(v.numpy(), g["tf_verts_out"]) < TOL and rel_err(j.numpy(), g["tf_joints_out"]) < TOL
    rr = geometry_ref.batch_rodrigues_quat(torch.from_numpy(g["rodrigues_in"]))
    assert np.allclose(rr.numpy(), g["rodrigues_out"], atol=1e-6)


def test_ief_known_answers(copenet_sd):
    """zeroed decoders return the init; swapping the views swaps the outputs."""
    torch.manual_seed(3)
    B = 3
    xf0, xf1 = torch.randn(B, 2048), torch.randn(B, 2048)
    bb0, bb1 = torch.rand(B, 3), torch.rand(B, 3)
    pos0, pos1 = torch.randn(B, 3), torch.randn(B, 3)
    with torch.no_grad():
        a = copenet_ref.ief(copenet_sd, xf0, xf1, bb0, bb1, pos0, pos1, iters=3)
        b = copenet_ref.ief(copenet_sd, xf1, xf0, bb1, bb0, pos1, pos0, iters=3)
        for x, y in zip(a, (b[2], b[3], b[0], b[1])):
            assert torch.equal(x, y)
        sdz = dict(copenet_sd)
        for k in ("decpose", "decshape"):
            sdz[k + ".weight"] = torch.zeros_like(sdz[k + ".weight"])
            sdz[k + ".bias"] = torch.zeros_like(sdz[k + ".bias"])
        p0, s0, p1, s1 = copenet_ref.ief(sdz, xf0, xf1, bb0, bb1, pos0, pos1, iters=3)
    assert torch.allclose(p0[:, :3], pos0) and torch.allclose(s1, copenet_sd["init_shape"].expand(B, -1))
    assert torch.allclose(p0[:, 3:], copenet_sd["init_pose"][:, :132].expand(B, -1))


def test_oracle_angle_axis_known_answers():
    """rotation_matrix_to_angle_axis (torchgeometry 0.1.2 restatement, parity unpinned): inverse of the axis-angle ->
    rotation map for angles below pi, identity -> 0, and the caller's zero-padded (N,3,4) input form."""
    import torch
    from oracle import geometry_ref as G
    g = torch.Generator().manual_seed(11)
    axis = torch.randn(500, 3, generator=g, dtype=torch.float64)
    axis = axis / axis.norm(dim=1, keepdim=True)
    ang = torch.rand(500, 1, generator=g, dtype=torch.float64) * 3.1
    aa = axis * ang
    K = torch.zeros(500, 3, 3, dtype=torch.float64)
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 0], K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -axis[:, 2], axis[:, 1], axis[:, 2], -axis[:, 0], -axis[:, 1], axis[:, 0]
    R = torch.eye(3, dtype=torch.float64) + torch.sin(ang).view(-1, 1, 1) * K + (1 - torch.cos(ang)).view(-1, 1, 1) * (K @ K)
    got = G.rotation_matrix_to_angle_axis(torch.cat([R, torch.zeros(500, 3, 1, dtype=torch.float64)], 2))
    assert (got - aa).abs().max() < 1e-9
    assert G.rotation_matrix_to_angle_axis(torch.eye(3, dtype=torch.float64).unsqueeze(0)).abs().max() == 0
    # 180 degrees about x (the mean pose's root joint): |angle| = pi, axis +-x
    Rx = torch.diag(torch.tensor([1.0, -1.0, -1.0], dtype=torch.float64)).unsqueeze(0)
    a = G.rotation_matrix_to_angle_axis(Rx)
    assert abs(abs(a[0, 0].item()) - 3.141592653589793) < 1e-9 and a[0, 1:].abs().max() < 1e-9


def test_oracle_copenet_sep_matches_reference(golden):
    """Two-weight-set model (copenet_real/models/model_copenet_sep.py): the oracle's sep IEF from the golden trunk
    features equals the imported reference's forward (forward_feat_ext patched to those features)."""
    import torch
    from airpose_amd import weights as W
    from oracle import copenet_ref
    from conftest import MEAN_PARAMS
    g, gs = golden["copenet_b2"], golden["copenet_sep_b2"]
    sd0 = W.to_torch(W.copenet_state_dict(int(gs["weights_seed0"]), MEAN_PARAMS))
    sd1 = W.to_torch(W.copenet_state_dict(int(gs["weights_seed1"]), MEAN_PARAMS))
    t = lambda a: torch.from_numpy(a)
    pos = t(g["init_position"])
    for it in (1, 3):
        out = copenet_ref.sep_ief(sd0, sd1, t(g["xf0"]), t(g["xf1"]), t(g["bb0"]), t(g["bb1"]), pos, pos, iters=it)
        for got, key in zip(out, ("pose0", "betas0", "pose1", "betas1")):
            assert rel_err(got.numpy(), gs["%s_it%d" % (key, it)]) < 2e-6, (key, it)


def test_oracle_singleview_matches_reference(golden):
    """copenet_singleview baseline (models/model_copenet_singleview.py): oracle vs the imported reference's forward."""
    import torch
    from airpose_amd import weights as W
    from oracle import copenet_ref
    g = golden["singleview_b1"]
    sd = W.to_torch(W.copenet_state_dict(int(g["weights_seed"]), MEAN_PARAMS, variant="singleview"))
    assert sorted(sd.keys()) == sorted(str(k) for k in g["state_dict_keys"])
    inp = W.synthetic_inputs(int(g["inputs_seed"]), 1)
    with torch.no_grad():
        pose, betas = copenet_ref.singleview_forward(sd, torch.from_numpy(inp["im0"]), torch.from_numpy(inp["bb0"]),
                                                     torch.from_numpy(g["init_position"]), iters=3)
    assert pose_err(pose.numpy(), g["pose"]) < 2e-6 and rel_err(betas.numpy(), g["betas"]) < 2e-6


def test_oracle_muhmr_matches_reference(golden):
    """muhmr two-view baseline (models/model_muhmr.py): oracle vs the imported reference's forward."""
    import torch
    from airpose_amd import weights as W
    from oracle import copenet_ref
    g = golden["muhmr_b1"]
    sd = W.to_torch(W.copenet_state_dict(int(g["weights_seed"]), MEAN_PARAMS, variant="muhmr"))
    assert sorted(sd.keys()) == sorted(str(k) for k in g["state_dict_keys"])
    inp = W.synthetic_inputs(int(g["inputs_seed"]), 1)
    with torch.no_grad():
        out = copenet_ref.muhmr_forward(sd, torch.from_numpy(inp["im0"]), torch.from_numpy(inp["im1"]), iters=3)
    for got, key in zip(out, ("pose0", "betas0", "cam0", "pose1", "betas1", "cam1")):
        assert rel_err(got.numpy(), g[key]) < 2e-6, key


def test_oracle_preprocess_known_answers():
    """Input pipeline restatement (parity unpinned: cv2 absent): identity size = pixel copy, constant stays constant,
    the bilinear core equals torch's independent implementation of the same half-pixel / clamp rule."""
    import torch
    import torch.nn.functional as F
    from oracle import preprocess_ref as P
    rs = np.random.RandomState(5)
    img = rs.rand(37, 61, 3)
    assert np.array_equal(P.cv2_resize_linear(img, 61, 37), img)
    assert np.allclose(P.cv2_resize_linear(np.full((20, 30, 3), 0.25), 77, 51), 0.25, atol=1e-12)
    for (dw, dh) in ((224, 136), (100, 33), (122, 75)):
        want = F.interpolate(torch.from_numpy(img).permute(2, 0, 1)[None], size=(dh, dw), mode="bilinear", align_corners=False)
        got = P.cv2_resize_linear(img, dw, dh)
        assert np.abs(got - want[0].permute(1, 2, 0).numpy()).max() < 1e-5      # (float32 taps in cv2's algorithm: ~4e-6 at source coordinate 60)
    frame = (rs.rand(120, 200, 3) * 255).astype(np.uint8)
    out, scale, pad = P.preprocess(frame, (10, 110, 30, 80))            # 100 x 50 crop -> 224 x 112, padded left/right
    assert scale == 2.24 and pad == [56, 0] and out.shape == (3, 224, 224)
    assert np.allclose(out[:, :, :56], (-P.MEAN / P.STD)[:, None, None], atol=1e-6)
    assert np.allclose(out[0, 0, 56] * P.STD[0] + P.MEAN[0], frame[10, 30, 2] / 255.0, atol=1e-6)   # BGR -> RGB, corner clamp
