"""ORACLE (test infrastructure only).  CPU restatement of the geometry helpers on the hot path.

  rot6d_to_rotmat          copenet/src/copenet/utils/geometry.py:47-61
  perspective_projection   copenet/src/copenet/utils/geometry.py:63-91
  transform_smpl           copenet/src/copenet/utils/utils.py:237-256
  batch_rodrigues (quat)   copenet/src/copenet/utils/geometry.py:9-45   (dataset GT only; off the inference path)

PINNED: tests/test_oracle_golden.py compares each function with tests/golden/geometry.npz,
produced by tools/make_golden.py from the imported reference modules.
"""
import torch


def rot6d_to_rotmat(x):
    """geometry.py:47-61.  The six numbers of a joint are a row-major 3x2 (a1 = even, a2 = odd entries)."""
    x = x.reshape(-1, 3, 2)
    a1, a2 = x[:, :, 0], x[:, :, 1]
    # F.normalize: v / max(||v||_2, 1e-12)
    b1 = a1 / a1.norm(dim=1, keepdim=True).clamp_min(1e-12)
    u2 = a2 - (b1 * a2).sum(dim=1, keepdim=True) * b1
    b2 = u2 / u2.norm(dim=1, keepdim=True).clamp_min(1e-12)
    b3 = torch.linalg.cross(b1, b2, dim=1)
    return torch.stack((b1, b2, b3), dim=-1)          # columns


def perspective_projection(points, rotation, translation, focal_length, camera_center):
    """geometry.py:63-91.  camera_center may be (B,2) or the caller's (1,B,2) (copenet_twoview.py:311)."""
    B = points.shape[0]
    K = torch.zeros(B, 3, 3, dtype=points.dtype)
    K[:, 0, 0] = focal_length[0]
    K[:, 1, 1] = focal_length[1]
    K[:, 2, 2] = 1.0
    K[:, :-1, -1] = camera_center
    p = torch.einsum("bij,bkj->bki", rotation, points) + translation.unsqueeze(1)
    p = p / p[:, :, -1].unsqueeze(-1)
    p = torch.einsum("bij,bkj->bki", K, p)
    return p[:, :, :-1]


def transform_smpl(trans_mat, vertices, joints=None):
    """utils.py:237-256 (vertices/joints legs): X' = R X + t, rotation about the ORIGIN."""
    R, t = trans_mat[:, :3, :3], trans_mat[:, :3, 3]
    v = torch.bmm(R, vertices.permute(0, 2, 1)).permute(0, 2, 1) + t.unsqueeze(1)
    j = None
    if joints is not None:
        j = torch.bmm(R, joints.permute(0, 2, 1)).permute(0, 2, 1) + t.unsqueeze(1)
    return v, j


def batch_rodrigues_quat(theta):
    """geometry.py:9-45: axis-angle -> rotmat through a unit quaternion."""
    n = torch.norm(theta + 1e-8, p=2, dim=1, keepdim=True)
    axis = theta / n
    half = n * 0.5
    q = torch.cat([torch.cos(half), torch.sin(half) * axis], dim=1)
    q = q / q.norm(p=2, dim=1, keepdim=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return torch.stack([
        w * w + x * x - y * y - z * z, 2 * x * y - 2 * w * z, 2 * w * y + 2 * x * z,
        2 * w * z + 2 * x * y, w * w - x * x + y * y - z * z, 2 * y * z - 2 * w * x,
        2 * x * z - 2 * w * y, 2 * w * x + 2 * y * z, w * w - x * x - y * y + z * z], dim=1).view(-1, 3, 3)
