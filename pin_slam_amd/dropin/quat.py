"""Quaternion helpers (w, x, y, z) of NeuralPoints.adjust_map, restating utils/tools.py:441-456 / 499-514."""
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
    """Batched rotation matrix -> quaternion, the reference's formula (w from the trace, no normalisation)."""
    qw = torch.sqrt(1.0 + R[:, 0, 0] + R[:, 1, 1] + R[:, 2, 2]) / 2.0
    qx = (R[:, 2, 1] - R[:, 1, 2]) / (4.0 * qw)
    qy = (R[:, 0, 2] - R[:, 2, 0]) / (4.0 * qw)
    qz = (R[:, 1, 0] - R[:, 0, 1]) / (4.0 * qw)
    return torch.stack((qw, qx, qy, qz), dim=1)
