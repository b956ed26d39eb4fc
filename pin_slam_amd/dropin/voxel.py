"""voxel_down_sample_min_value (utils/tools.py:629-668) for NeuralPoints.recreate_hash:
per voxel the index of the point with the smallest `value` (quantised to 1000 levels, lowest
index on ties), ordered by voxel id.  Next-tier row: host-side torch ops."""
import torch


def voxel_down_sample_min_value(points: torch.Tensor, voxel_size: float, value: torch.Tensor) -> torch.Tensor:
    offset = torch.floor(points.min(dim=0)[0] / voxel_size).long()
    grid = torch.floor(points / voxel_size).long() - offset
    v = grid.max()
    gid = grid[:, 0] + grid[:, 1] * v + grid[:, 2] * v * v
    _, inverse = torch.unique(gid, return_inverse=True)
    n = inverse.size(0)
    idx = torch.arange(n, dtype=inverse.dtype, device=inverse.device)
    off = 10 ** len(str(n - 1))
    q = (value / value.max() * 999).long()
    key = idx + q * off
    out = torch.empty(int(inverse.max().item()) + 1, dtype=inverse.dtype, device=inverse.device)
    out.scatter_reduce_(0, inverse, key, reduce="amin", include_self=False)
    return out % off
