"""Quaternion helpers (w, x, y, z) of NeuralPoints.adjust_map, restating utils/tools.py:441-456 (the quaternion product itself runs in pin_transform_by_frame)."""
import torch


def rotmat_to_quat(R: torch.Tensor) -> torch.Tensor:
    """Batched rotation matrix -> quaternion, the reference's formula (w from the trace, no normalisation)."""
    qw = torch.sqrt(1.0 + R[:, 0, 0] + R[:, 1, 1] + R[:, 2, 2]) / 2.0
    qx = (R[:, 2, 1] - R[:, 1, 2]) / (4.0 * qw)
    qy = (R[:, 0, 2] - R[:, 2, 0]) / (4.0 * qw)
    qz = (R[:, 1, 0] - R[:, 0, 1]) / (4.0 * qw)
    return torch.stack((qw, qx, qy, qz), dim=1)
