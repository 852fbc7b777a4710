"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every symbol
include/airpose_hip.h declares (no compute calls: there is no GPU here), the host mirrors keep the
reference's contracts, and the product refuses to run without the GPU instead of falling back."""
import os
import ctypes
import re

import numpy as np
import pytest
import torch

from conftest import MEAN_PARAMS, REPO


def _declared():
    src = open(os.path.join(REPO, "include", "airpose_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ap_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from airpose_amd import _native
    if not os.path.isfile(_native.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    L = _native.lib()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), n
        assert n in _native.SIGNATURES, "ctypes signature missing for " + n
    assert sorted(_native.SIGNATURES) == names
    assert b"gfx950" in L.ap_version()


def test_abi_version_matches_header_and_binding():
    """An out-of-tree caller built against another header must be able to tell: the library exports the ABI number of the header
    it was built from, and the ctypes binding refuses any other."""
    import re
    from airpose_amd import _native
    hdr = open(os.path.join(REPO, "include", "airpose_hip.h")).read()
    want = int(re.search(r"#define\s+AP_ABI_VERSION\s+(\d+)", hdr).group(1))
    L = ctypes.CDLL(_native.LIB_PATH)
    L.ap_abi_version.restype = ctypes.c_int
    assert L.ap_abi_version() == want == _native.ABI_VERSION
    L.ap_version.restype = ctypes.c_char_p
    assert ("abi %d" % want).encode() in L.ap_version()


def test_state_dict_contract_matches_reference(golden, copenet_sd):
    from airpose_amd import copenet_model
    net = copenet_model.getcopenet(MEAN_PARAMS)
    ref_keys = [str(k) for k in golden["copenet_b2"]["state_dict_keys"]]
    assert list(net.state_dict().keys()) == ref_keys            # same names, same order, 331 entries
    res = net.load_state_dict(copenet_sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert net.init_pose.shape == (1, 144) and net.init_shape.shape == (1, 10) and net.init_cam.shape == (1, 3)
    for attr in ("fc1", "fc2", "decpose", "decshape", "deccam", "conv1", "bn1", "layer1", "layer4"):
        assert hasattr(net, attr)
    assert net.fc1.in_features == 2332 and net.decpose.out_features == 135


def test_lightning_style_checkpoint_loads(copenet_sd, tmp_path):
    """Lightning checkpoints prefix the network with 'model.' (copenet_twoview.py:60)."""
    from airpose_amd import copenet_model

    class Wrapper(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.model = copenet_model.getcopenet(MEAN_PARAMS)

    ck = {"state_dict": {"model." + k: v for k, v in copenet_sd.items()}}
    p = str(tmp_path / "last.ckpt")
    torch.save(ck, p)
    w = Wrapper()
    w.load_state_dict(torch.load(p)["state_dict"], strict=True)
    assert torch.equal(w.model.fc1.weight, copenet_sd["fc1.weight"])


def test_no_cpu_fallback(copenet_sd):
    from airpose_amd import copenet_model, geometry, smplx, smplx_model
    net = copenet_model.getcopenet(MEAN_PARAMS).eval()
    with pytest.raises(RuntimeError, match="no CPU"):
        net(torch.zeros(1, 3, 224, 224), torch.zeros(1, 3, 224, 224), torch.zeros(1, 3), torch.zeros(1, 3),
            torch.zeros(1, 3), torch.zeros(1, 3))
    with pytest.raises(RuntimeError, match="no CPU"):
        geometry.rot6d_to_rotmat(torch.zeros(2, 6))
    small = smplx_model.make_synthetic_model(1, num_verts=300, num_faces=100)
    body = smplx.SMPLX(model_data=small)
    with pytest.raises(RuntimeError, match="no CPU"):
        body.forward(betas=torch.zeros(1, 10), body_pose=torch.eye(3).expand(1, 21, 3, 3), pose2rot=False)
    with pytest.raises(FileNotFoundError):
        smplx.SMPLX("/nonexistent/models/smplx")


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under airpose_amd/ may import or call it."""
    pkg = os.path.join(REPO, "airpose_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "/root/reference" not in txt, f


def test_synthetic_inputs_contract():
    from airpose_amd import weights as W
    d = W.synthetic_inputs(1234, 3)
    assert d["im0"].shape == (3, 3, 224, 224) and d["im0"].dtype == np.float32
    assert d["bb0"].shape == (3, 3) and (d["bb0"][:, 2] >= 0.2).all()
    assert d["intr0"][0, 0, 0] == 1475 and d["intr0"][0, 0, 2] == 960 and d["intr0"][0, 1, 2] == 540


def test_conv_config_knob_validates_on_the_host():
    """ap_set_conv_config is host-only state: every documented value is accepted, anything else is refused with AP_EINVAL
    and leaves the automatic choice in place (no GPU needed)."""
    from airpose_amd import _native as Nn
    L = Nn.lib()
    ok = [-1, -4, -5, 100, 17] + list(range(0, 15))
    try:
        for c in ok:
            assert L.ap_set_conv_config(c) == 0, c
        for c in (-2, -3, -6, 15, 16, 18, 19, 20, 24, 28, 99, 101):
            assert L.ap_set_conv_config(c) != 0, c
    finally:
        assert L.ap_set_conv_config(-1) == 0


def test_one_library_carries_both_16bit_storage_types():
    """AP_PREC_F16 is a precision of the ONE library (no second .so, no flavour switch): the enum values are distinct, the
    f16 kernel set (namespace k_f16) is linked in, and the host-only entry points validate the precision argument."""
    import subprocess
    from airpose_amd import _native as Nn
    assert sorted(Nn.PRECISIONS.values()) == [0, 1, 2, 3] and Nn.PRECISIONS["f16"] == 3 and Nn.PRECISIONS["bf16"] == 1
    assert not os.path.exists(os.path.join(REPO, "airpose_amd", "libairpose_hip_f16.so"))
    syms = subprocess.run(["nm", "-D", "--defined-only", Nn.LIB_PATH], check=True, capture_output=True, text=True).stdout
    for ns in ("k_bf16", "k_f16"):
        for fn in ("ap_launch_conv_pipe", "ap_launch_conv_pair", "ap_launch_bneck2", "ap_launch_stem_pool", "ap_launch_conv_slab"):
            assert re.search(r"_ZN%d%s\d+%s" % (len(ns), ns, fn), syms), (ns, fn)
    L = Nn.lib()
    # pair-stream size: a host-only query; unsupported shapes are refused
    assert L.ap_conv_pair_stream_bytes(128, 0, 128) > 0 and L.ap_conv_pair_stream_bytes(256, 512, 0) > 0
    assert L.ap_conv_pair_stream_bytes(64, 0, 64) < 0
    # the stand-alone 16-bit operators take AP_PREC_BF16 / AP_PREC_F16 only (argument check comes before any launch)
    assert L.ap_conv_pair_pack(Nn.AP_PREC_FP32, None, None, 128, 0, 128, None, None) == -1
    assert L.ap_net_range_status(None, None, 0) == -1 and L.ap_net_set_range_check(None, 1) == -1


def test_bench_launches_itself_for_n_gpus():
    """`python bench.py --gpus 2` with no launcher environment (the shape of the driver's plain command) must start its own ranks
    (it used to die with SystemExit before touching a GPU).  AIRPOSE_BENCH_LAUNCH_ONLY=1 stops each rank where the GPU
    would be needed: self-launch through torch.distributed.run, gloo rendezvous on 127.0.0.1, one all_reduce, ONE JSON line."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["AIRPOSE_BENCH_LAUNCH_ONLY"] = "1"
    env["OMP_NUM_THREADS"] = "1"
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["launcher"] == "ok" and rec["n_gpus"] == 2 and rec["n_ranks_seen"] == 2
    # and a launcher that started another number of ranks than --gpus names is refused loudly
    env2 = dict(env, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    bad = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2"], env=env2, capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "WORLD_SIZE=3" in (bad.stderr + bad.stdout)
