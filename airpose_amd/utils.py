"""GPU version of ``transform_smpl`` (copenet/src/copenet/utils/utils.py:237-256): X' = R X + t about the origin."""
import torch

from . import _native as N


def _apply(rt, pts):
    dev = pts.device
    out = torch.empty_like(pts, dtype=torch.float32)
    with torch.cuda.device(dev):
        N.check(N.lib().ap_transform_points(N.dptr(rt), N.dptr(pts), pts.shape[0], pts.shape[1], N.dptr(out),
                                            N.stream_ptr(dev)), "ap_transform_points")
    return out


def transform_smpl(trans_mat, smplvertices=None, smpljoints=None, orientation=None, smpltrans=None):
    """Returns (verts, joints, orient, trans) like the reference; trans_mat (B,3,4) or (B,4,4)."""
    if not trans_mat.is_cuda:
        raise RuntimeError("airpose_amd.utils.transform_smpl: CUDA (ROCm) tensors only; there is no CPU path")
    rt = N.f32c(trans_mat[:, :3, :4])
    verts = _apply(rt, N.f32c(smplvertices)) if smplvertices is not None else None
    joints = _apply(rt, N.f32c(smpljoints)) if smpljoints is not None else None
    trans = _apply(rt, N.f32c(smpltrans).unsqueeze(1)).squeeze(1) if smpltrans is not None else None
    orient = None
    if orientation is not None:
        # R @ orientation: the 3 columns of `orientation` are 3 points rotated without translation
        r0 = torch.cat([rt[:, :, :3], torch.zeros_like(rt[:, :, 3:])], dim=2).contiguous()
        cols = N.f32c(orientation).transpose(1, 2).contiguous()
        orient = _apply(r0, cols).transpose(1, 2).contiguous()
    return verts, joints, orient, trans
