"""GPU versions of the reference's geometry helpers on the hot path
(copenet/src/copenet/utils/geometry.py:47-61 rot6d_to_rotmat, :63-91 perspective_projection).
Each is one launch of a hand-written HIP kernel through the C ABI; no CPU path."""
import torch

from . import _native as N


def _cuda(t, name):
    if not t.is_cuda:
        raise RuntimeError("airpose_amd.geometry.%s: CUDA (ROCm) tensors only; there is no CPU path" % name)
    return t.device


def rot6d_to_rotmat(x):
    """(B,6k) 6-D rotations -> (B*k,3,3)   [geometry.py:47-61]"""
    dev = _cuda(x, "rot6d_to_rotmat")
    x = N.f32c(x).reshape(-1, 6)
    out = torch.empty(x.shape[0], 3, 3, device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        N.check(N.lib().ap_rot6d_to_rotmat(N.dptr(x), x.shape[0], N.dptr(out), N.stream_ptr(dev)), "ap_rot6d_to_rotmat")
    return out


def _rodrigues(theta, variant, name):
    dev = _cuda(theta, name)
    t = N.f32c(theta).reshape(-1, 3)
    out = torch.empty(t.shape[0], 3, 3, device=dev, dtype=torch.float32)
    if t.shape[0]:
        with torch.cuda.device(dev):
            N.check(N.lib().ap_batch_rodrigues(N.dptr(t), t.shape[0], variant, N.dptr(out), N.stream_ptr(dev)),
                    "ap_batch_rodrigues")
    return out


def batch_rodrigues(theta):
    """(N,3) axis-angle -> (N,3,3) through a unit quaternion   [geometry.py:9-45]"""
    return _rodrigues(theta, 1, "batch_rodrigues")


def rotation_matrix_to_angle_axis(rotation_matrix):
    """tgm.rotation_matrix_to_angle_axis (torchgeometry 0.1.2): (N,3,4) -- or (N,3,3) -- -> (N,3), the conversion the
    caller applies to pred_rotmat for pred_angles [copenet_twoview.py:323-324]."""
    dev = _cuda(rotation_matrix, "rotation_matrix_to_angle_axis")
    r = N.f32c(rotation_matrix)
    if r.dim() != 3 or r.shape[1] != 3 or r.shape[2] not in (3, 4):
        raise ValueError("rotation_matrix_to_angle_axis: expected (N,3,4) or (N,3,3), got %s" % (tuple(r.shape),))
    out = torch.empty(r.shape[0], 3, device=dev, dtype=torch.float32)
    if r.shape[0]:
        with torch.cuda.device(dev):
            N.check(N.lib().ap_rotmat_to_angle_axis(N.dptr(r), r.shape[0], r.shape[2], N.dptr(out), N.stream_ptr(dev)),
                    "ap_rotmat_to_angle_axis")
    return out


def perspective_projection(points, rotation, translation, focal_length, camera_center):
    """(bs,N,3) -> (bs,N,2)   [geometry.py:63-91]; camera_center (bs,2) or the caller's (1,bs,2)."""
    dev = _cuda(points, "perspective_projection")
    B, P = points.shape[0], points.shape[1]
    points = N.f32c(points)
    rotation = N.f32c(rotation, dev)
    translation = N.f32c(translation, dev)
    cc = N.f32c(camera_center, dev).reshape(-1, 2)
    if cc.shape[0] != B:
        cc = cc.expand(B, 2).contiguous()
    out = torch.empty(B, P, 2, device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        N.check(N.lib().ap_perspective_projection(N.dptr(points), B, P, N.dptr(rotation), N.dptr(translation),
                                                  float(focal_length[0]), float(focal_length[1]), N.dptr(cc),
                                                  N.dptr(out), N.stream_ptr(dev)), "ap_perspective_projection")
    return out
