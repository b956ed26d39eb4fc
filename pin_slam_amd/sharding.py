"""Data-parallel sharding of the mapper batch: contiguous shards of the global
batch, Eikonal sub-sample = global indices i with i % dec == 0, loss normalised by GLOBAL
counts so that the SUM of the per-rank gradients equals the single-GPU gradient."""
from __future__ import annotations


def shard_range(bs_global: int, rank: int, world: int):
    """[start, stop) of rank's contiguous shard (bs_global must divide evenly)."""
    if bs_global % world:
        raise ValueError("global batch must be divisible by the world size")
    n = bs_global // world
    return rank * n, (rank + 1) * n


def eikonal_shard(start: int, n_local: int, dec: int):
    """(first, count): local offsets first + s*dec, s < count, are exactly the global indices
    in [start, start + n_local) that are multiples of dec."""
    first = (-start) % dec
    count = 0 if first >= n_local else (n_local - first + dec - 1) // dec
    return first, count


def n_eik_global(bs_global: int, dec: int) -> int:
    return (bs_global + dec - 1) // dec
