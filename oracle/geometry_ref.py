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


def rotation_matrix_to_quaternion(rotation_matrix, eps=1e-6):
    """torchgeometry==0.1.2 (pin: copenet/requirements.txt:16; the package is ABSENT here -> PARITY UNPINNED, held by
    known-answer tests only) `core/conversions.py::rotation_matrix_to_quaternion`, restated from the published source:
    (N,3,4) -> (N,4) [w, x, y, z]; branch masks on the TRANSPOSED matrix, exactly as published.
    Call sites: copenet_twoview.py:323-326 via rotation_matrix_to_angle_axis."""
    rt = rotation_matrix[:, :3, :3].transpose(1, 2)
    m_d2 = rt[:, 2, 2] < eps
    m_d0_d1 = rt[:, 0, 0] > rt[:, 1, 1]
    m_d0_nd1 = rt[:, 0, 0] < -rt[:, 1, 1]
    t0 = 1 + rt[:, 0, 0] - rt[:, 1, 1] - rt[:, 2, 2]
    q0 = torch.stack([rt[:, 1, 2] - rt[:, 2, 1], t0, rt[:, 0, 1] + rt[:, 1, 0], rt[:, 2, 0] + rt[:, 0, 2]], -1)
    t1 = 1 - rt[:, 0, 0] + rt[:, 1, 1] - rt[:, 2, 2]
    q1 = torch.stack([rt[:, 2, 0] - rt[:, 0, 2], rt[:, 0, 1] + rt[:, 1, 0], t1, rt[:, 1, 2] + rt[:, 2, 1]], -1)
    t2 = 1 - rt[:, 0, 0] - rt[:, 1, 1] + rt[:, 2, 2]
    q2 = torch.stack([rt[:, 0, 1] - rt[:, 1, 0], rt[:, 2, 0] + rt[:, 0, 2], rt[:, 1, 2] + rt[:, 2, 1], t2], -1)
    t3 = 1 + rt[:, 0, 0] + rt[:, 1, 1] + rt[:, 2, 2]
    q3 = torch.stack([t3, rt[:, 1, 2] - rt[:, 2, 1], rt[:, 2, 0] - rt[:, 0, 2], rt[:, 0, 1] - rt[:, 1, 0]], -1)
    c0 = (m_d2 & m_d0_d1).view(-1, 1).type_as(q0)
    c1 = (m_d2 & ~m_d0_d1).view(-1, 1).type_as(q0)
    c2 = (~m_d2 & m_d0_nd1).view(-1, 1).type_as(q0)
    c3 = (~m_d2 & ~m_d0_nd1).view(-1, 1).type_as(q0)
    q = q0 * c0 + q1 * c1 + q2 * c2 + q3 * c3
    q = q / torch.sqrt(t0.view(-1, 1) * c0 + t1.view(-1, 1) * c1 + t2.view(-1, 1) * c2 + t3.view(-1, 1) * c3)
    return q * 0.5


def quaternion_to_angle_axis(quaternion):
    """torchgeometry==0.1.2 `quaternion_to_angle_axis`: (N,4) [w,x,y,z] -> (N,3)."""
    q1, q2, q3 = quaternion[..., 1], quaternion[..., 2], quaternion[..., 3]
    sin_sq = q1 * q1 + q2 * q2 + q3 * q3
    sin_t = torch.sqrt(sin_sq)
    cos_t = quaternion[..., 0]
    two_theta = 2.0 * torch.where(cos_t < 0.0, torch.atan2(-sin_t, -cos_t), torch.atan2(sin_t, cos_t))
    k = torch.where(sin_sq > 0.0, two_theta / sin_t, 2.0 * torch.ones_like(sin_t))
    return torch.stack([q1 * k, q2 * k, q3 * k], -1)


def rotation_matrix_to_angle_axis(rotation_matrix):
    """torchgeometry==0.1.2 `rotation_matrix_to_angle_axis` ((N,3,4) or (N,3,3) -> (N,3)); copenet_twoview.py:323-324."""
    return quaternion_to_angle_axis(rotation_matrix_to_quaternion(rotation_matrix))
