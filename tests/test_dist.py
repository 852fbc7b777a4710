"""Multi-process (world size 2, gloo, CPU) tests of the N>1 host logic: pair sharding and the view-split
IEF exchange.  The per-view step function is the oracle here (no GPU in this container); on the GPU box
the same driver runs ap_regressor_step (tests/test_gpu_parity.py checks that step against the oracle)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import MEAN_PARAMS, REPO


def test_shard_pairs_cover_and_balance():
    from airpose_amd.dist import shard_pairs
    for n, w in ((2048, 8), (10, 4), (7, 8), (256, 1)):
        spans = [shard_pairs(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


def test_wire_format_roundtrip():
    from airpose_amd.dist import pack_wire, unpack_wire
    pose, betas = torch.randn(3, 135), torch.randn(3, 10)
    msg = pack_wire(pose, betas)
    assert msg.shape == (3, 145)
    assert torch.equal(msg[:, :10], betas) and torch.equal(msg[:, 10:13], pose[:, :3])
    p2, b2 = unpack_wire(msg)
    assert torch.equal(p2, pose) and torch.equal(b2, betas)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from airpose_amd import dist as D
    from airpose_amd import weights as W
    from oracle import copenet_ref
    sd = W.to_torch(W.copenet_state_dict(20240901, MEAN_PARAMS))
    grp, v = rank // 2, rank % 2                               # pair group and this rank's view
    g = torch.Generator().manual_seed(17 + grp)                # same data on both ranks of a pair group, other data per group
    B = 3
    xf = [torch.randn(B, 2048, generator=g), torch.randn(B, 2048, generator=g)]
    bb = [torch.rand(B, 3, generator=g), torch.rand(B, 3, generator=g)]
    pos = [torch.randn(B, 3, generator=g) * 0.3, torch.randn(B, 3, generator=g) * 0.3]

    def step(xf_v, bb_v, pose, betas, partner):
        # this view's half of forward_reg (model_copenet.py:185-188,198-199) from the oracle's linear layers
        xc = torch.cat([xf_v, bb_v, pose[:, :3], pose[:, 3:9], pose[:, 9:], betas, partner[:, :126], partner[:, 126:]], 1)
        h = copenet_ref._lin(copenet_ref._lin(xc, sd, "fc1"), sd, "fc2")
        return pose + copenet_ref._lin(h, sd, "decpose"), betas + copenet_ref._lin(h, sd, "decshape")

    # the same step in its partner-independent and partner-dependent halves (fc1 -> fc2 -> dec is affine: model_copenet.py:186-202):
    # delta(xc) = dec(fc2(fc1(xc))); local = delta([xf | bb | state | 0]), finish adds delta([0 | 0 | 0 | partner]) - delta(0)
    def delta(xc):
        h = copenet_ref._lin(copenet_ref._lin(xc, sd, "fc1"), sd, "fc2")
        return torch.cat([copenet_ref._lin(h, sd, "decpose"), copenet_ref._lin(h, sd, "decshape")], 1)

    def step_local(hfeat, bb_v, pose, betas):
        return delta(torch.cat([hfeat, bb_v, pose, betas, torch.zeros(pose.shape[0], 136)], 1))

    def step_finish(partial, pose, betas, partner):
        z = torch.zeros(pose.shape[0], 2332)
        zp = z.clone()
        zp[:, 2196:] = partner
        d = partial + (delta(zp) - delta(z))
        return pose + d[:, :135], betas + d[:, 135:]

    groups = D.make_pair_groups(world)
    ief = D.ViewSplitIEF(step, groups[rank // 2], (2 * (rank // 2), 2 * (rank // 2) + 1))
    ief_ov = D.ViewSplitIEF(step, groups[rank // 2], (2 * (rank // 2), 2 * (rank // 2) + 1),
                            split_step=(lambda xf_v: xf_v, step_local, step_finish))
    with torch.no_grad():
        pose, betas = ief.run(xf[v], bb[v], pos[v], sd["init_pose"], sd["init_shape"], iters=3, shared_init=True)
        assert ief.n_exchanges == 2            # both views start from the model's mean state: none before iteration 1
        want = copenet_ref.ief(sd, xf[0], xf[1], bb[0], bb[1], pos[0], pos[1], iters=3)
        # caller-supplied per-view initial state (model_copenet.py:121-136): the first exchange is needed
        th = [torch.randn(B, 132, generator=g) * 0.3, torch.randn(B, 132, generator=g) * 0.3]
        sh = [torch.randn(B, 10, generator=g) * 0.3, torch.randn(B, 10, generator=g) * 0.3]
        pose_c, betas_c = ief.run(xf[v], bb[v], pos[v], th[v], sh[v], iters=2, shared_init=False)
        assert ief.n_exchanges == 4
        want_c = copenet_ref.ief(sd, xf[0], xf[1], bb[0], bb[1], pos[0], pos[1], init_theta0=th[0], init_theta1=th[1],
                                 init_shape0=sh[0], init_shape1=sh[1], iters=2)
        # the overlapped form (exchange issued before the partner-independent half of the step, waited for before the other half)
        pose_o, betas_o = ief_ov.run(xf[v], bb[v], pos[v], sd["init_pose"], sd["init_shape"], iters=3, shared_init=True)
        assert ief_ov.n_exchanges == 2
        pose_oc, betas_oc = ief_ov.run(xf[v], bb[v], pos[v], th[v], sh[v], iters=2, shared_init=False)
        assert ief_ov.n_exchanges == 4
        # the partner rows handed out by exchange() belong to the caller: a later exchange must not change them
        # (they used to be a view of the cached gather buffer)
        p1 = ief.exchange(pose, betas)
        keep = p1.clone()
        p2 = ief.exchange(pose + 1.0, betas + 1.0)
        assert torch.equal(p1, keep) and not torch.equal(p1, p2)
        assert p1.data_ptr() != p2.data_ptr()
    np.savez(os.path.join(out_dir, "r%d.npz" % rank), pose=pose.numpy(), betas=betas.numpy(),
             want_pose=want[2 * v].numpy(), want_betas=want[2 * v + 1].numpy(),
             pose_c=pose_c.numpy(), betas_c=betas_c.numpy(), pose_o=pose_o.numpy(), betas_o=betas_o.numpy(),
             pose_oc=pose_oc.numpy(), betas_oc=betas_oc.numpy(),
             want_pose_c=want_c[2 * v].numpy(), want_betas_c=want_c[2 * v + 1].numpy())
    # default sharding: no collective on the data path -- only the bench's barrier / max-reduce
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == world
    dist.barrier()
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("world", [2, 4])       # 4 ranks = two pair groups side by side (the first real multi-GPU run forms four)
def test_view_split_ief_matches_two_view_oracle(tmp_path, world):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    poses = []
    for r in range(world):
        d = np.load(str(tmp_path / ("r%d.npz" % r)))
        assert np.allclose(d["pose"], d["want_pose"], rtol=0, atol=2e-6)
        assert np.allclose(d["betas"], d["want_betas"], rtol=0, atol=2e-6)
        assert np.allclose(d["pose_c"], d["want_pose_c"], rtol=0, atol=2e-6)
        assert np.allclose(d["betas_c"], d["want_betas_c"], rtol=0, atol=2e-6)
        assert np.allclose(d["pose_o"], d["want_pose"], rtol=0, atol=2e-6) and np.allclose(d["betas_o"], d["want_betas"], rtol=0, atol=2e-6)
        assert np.allclose(d["pose_oc"], d["want_pose_c"], rtol=0, atol=2e-6) and np.allclose(d["betas_oc"], d["want_betas_c"], rtol=0, atol=2e-6)
        poses.append(d["pose"])
    if world == 4:                                            # the groups really carried different pairs
        assert not np.allclose(poses[0], poses[2])
