"""GPU version of ``transform_smpl`` (copenet/src/copenet/utils/utils.py:237-256): X' = R X + t about the origin."""
import ctypes

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


def preprocess_crops(frames, crops, bgr=True):
    """GPU version of the dataset's input pipeline (aerialpeople.py:125-141,174 + resize_with_pad, utils.py:214-235):
    uint8 HWC frames (n,H,W,3) -- or one (H,W,3) frame shared by all crops -- and crops (n,4) = [y0, y1, x0, x1]
    -> (images (n,3,224,224) float32 normalised, scale (n,), pad (n,2) = [pad_left, pad_top]).
    `bgr=True` for cv2.imread frames (the reference reverses the channel order)."""
    if not frames.is_cuda or frames.dtype != torch.uint8:
        raise RuntimeError("airpose_amd.utils.preprocess_crops: uint8 CUDA (ROCm) frames only; there is no CPU path")
    dev = frames.device
    crops = crops.to(device=dev, dtype=torch.int32).contiguous()
    n = crops.shape[0]
    shared = frames.dim() == 3
    if frames.shape[-1] != 3 or (not shared and (frames.dim() != 4 or frames.shape[0] != n)) or crops.shape != (n, 4):
        raise RuntimeError("preprocess_crops: frames (n,H,W,3) or (H,W,3) uint8, crops (n,4)")
    frames = frames.contiguous()
    H, W = frames.shape[-3], frames.shape[-2]
    c = crops.cpu()
    if n == 0 or (c[:, 0] < 0).any() or (c[:, 2] < 0).any() or (c[:, 1] > H).any() or (c[:, 3] > W).any() or \
            (c[:, 1] <= c[:, 0]).any() or (c[:, 3] <= c[:, 2]).any():
        raise RuntimeError("preprocess_crops: crops must be non-empty and inside the frame")
    out = torch.empty(n, 3, 224, 224, device=dev, dtype=torch.float32)
    scale = torch.empty(n, device=dev, dtype=torch.float32)
    pad = torch.empty(n, 2, device=dev, dtype=torch.int32)
    with torch.cuda.device(dev):
        raw = lambda t: ctypes.c_void_p(t.data_ptr())       # (uint8 / int32 buffers: N.dptr is for float32 tensors)
        N.check(N.lib().ap_preprocess_crops(raw(frames), 0 if shared else H * W * 3, n, H, W, int(bool(bgr)),
                                            raw(crops), N.dptr(out), N.dptr(scale), raw(pad), N.stream_ptr(dev)),
                "ap_preprocess_crops")
    return out, scale, pad
