"""Quaternion helpers (w,x,y,z) used by NeuralPoints.adjust_map -- restating the math of
utils/tools.py rotmat_to_quat / quat_multiply (next-tier, host-side torch)."""
import torch


def quat_multiply(q1: torch.Tensor, q2: torch.Tensor) -> torch.Tensor:
    w1, x1, y1, z1 = q1.unbind(-1)
    w2, x2, y2, z2 = q2.unbind(-1)
    return torch.stack([
        w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2,
        w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
        w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
        w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2], -1)


def rotmat_to_quat(R: torch.Tensor) -> torch.Tensor:
    """Batched rotation matrix -> unit quaternion (w >= 0 branch-free form)."""
    m00, m11, m22 = R[..., 0, 0], R[..., 1, 1], R[..., 2, 2]
    w = torch.sqrt(torch.clamp(1 + m00 + m11 + m22, min=1e-12)) / 2
    x = torch.sqrt(torch.clamp(1 + m00 - m11 - m22, min=0)) / 2
    y = torch.sqrt(torch.clamp(1 - m00 + m11 - m22, min=0)) / 2
    z = torch.sqrt(torch.clamp(1 - m00 - m11 + m22, min=0)) / 2
    x = torch.copysign(x, R[..., 2, 1] - R[..., 1, 2])
    y = torch.copysign(y, R[..., 0, 2] - R[..., 2, 0])
    z = torch.copysign(z, R[..., 1, 0] - R[..., 0, 1])
    q = torch.stack([w, x, y, z], -1)
    return q / q.norm(dim=-1, keepdim=True)
