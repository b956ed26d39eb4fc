"""Seeded synthetic workloads: wavy-sheet neural-point map, LiDAR-like
scan on the middle sheet, mapper sample pool.  Host-side numpy generation of *inputs* only --
nothing here is on the hot path, and nothing here touches the oracle or the reference."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

PRIMES = np.array([73856093, 19349669, 83492791], dtype=np.int64)


# sheet family z = z0 + spacing * layer + amp * sin(freq x) cos(freq y); SURVEY 8d: (-2, 0.8, 0.3, 0.5) for the LiDAR
# workloads C2-C4, sheets 0.1 m apart in a 10 m room for the RGB-D workload C5
DEFAULT_SHEETS = (-2.0, 0.8, 0.3, 0.5)


def sheet_height(x, y, layer, sheets=DEFAULT_SHEETS):
    z0, spacing, amp, freq = sheets
    return z0 + spacing * layer + amp * np.sin(freq * x) * np.cos(freq * y)


def sheet_normal(x, y, sheets=DEFAULT_SHEETS):
    _, _, amp, freq = sheets
    fx = amp * freq * np.cos(freq * x) * np.cos(freq * y)
    fy = -amp * freq * np.sin(freq * x) * np.sin(freq * y)
    n = np.stack([-fx, -fy, np.ones_like(fx)], -1)
    return n / np.linalg.norm(n, axis=-1, keepdims=True)


def disc_points(rng, n, radius, layers, center=(0.0, 0.0), sheets=DEFAULT_SHEETS):
    r = radius * np.sqrt(rng.random(n))
    th = 2 * np.pi * rng.random(n)
    x = center[0] + r * np.cos(th)
    y = center[1] + r * np.sin(th)
    layer = rng.integers(0, layers, n) if np.isscalar(layers) else rng.choice(layers, n)
    z = sheet_height(x, y, layer, sheets)
    return np.stack([x, y, z], 1).astype(np.float32), layer


@dataclass
class SynthMap:
    positions: np.ndarray   # [P,3] f32
    table: np.ndarray       # [B] int32
    features: np.ndarray    # [P+1,8] f32 (last row padding)
    resolution: float
    buffer_size: int
    layers: int
    radius: float
    sheets: tuple = DEFAULT_SHEETS


def hash_slots(cells, B):
    h = (cells.astype(np.int64) * PRIMES).sum(-1)
    return np.mod(h, np.int64(B))  # mathematical modulus == fmod + negative wrap


def build_map(layers=16, radius=80.0, resolution=0.4, buffer_size=int(5e7), raw_per_layer=1_600_000,
              feature_std=0.1, seed=0, sheets=DEFAULT_SHEETS) -> SynthMap:
    """One neural point per occupied voxel of `layers` wavy sheets in a disc (SURVEY 8d:
    L=4 -> ~5.6e5 points, L=16 -> ~2.2e6).  The kept point of a voxel is the first raw point
    that falls into it; the hash table is written in index order (last writer wins)."""
    rng = np.random.default_rng(seed)
    kept = []
    for l in range(layers):
        pts, _ = disc_points(rng, raw_per_layer, radius, [l], sheets=sheets)
        g = np.floor(pts / np.float32(resolution)).astype(np.int64)
        key = (g[:, 0] + 4096) + ((g[:, 1] + 4096) << 14) + ((g[:, 2] + 4096) << 28)
        _, first = np.unique(key, return_index=True)
        kept.append(pts[np.sort(first)])
    pos = np.concatenate(kept, 0)
    # merge voxels shared between layers (keep first)
    g = np.floor(pos / np.float32(resolution)).astype(np.int64)
    key = (g[:, 0] + 4096) + ((g[:, 1] + 4096) << 14) + ((g[:, 2] + 4096) << 28)
    _, first = np.unique(key, return_index=True)
    first = np.sort(first)
    pos, g = pos[first], g[first]
    table = np.full(buffer_size, -1, np.int32)
    table[hash_slots(g, buffer_size)] = np.arange(len(pos), dtype=np.int32)
    feats = (feature_std * rng.standard_normal((len(pos) + 1, 8))).astype(np.float32)
    return SynthMap(pos, table, feats, resolution, buffer_size, layers, radius, tuple(sheets))


def make_scan(m: SynthMap, n=100_000, noise=0.02, seed=1, radius=None):
    """LiDAR-like scan: points on the middle sheet + N(0, noise^2)."""
    rng = np.random.default_rng(seed)
    pts, _ = disc_points(rng, n, radius or m.radius * 0.95, [m.layers // 2], sheets=m.sheets)
    return (pts + noise * rng.standard_normal((n, 3))).astype(np.float32)


def make_pool(m: SynthMap, n=2_000_000, sigma=0.25, seed=2, radius=None):
    """Mapper sample pool: sheet points displaced along the normal by d ~ N(0, sigma^2),
    label = d (+ small noise), weight 1, ts 0."""
    rng = np.random.default_rng(seed)
    base, _ = disc_points(rng, n, radius or m.radius * 0.95, m.layers, sheets=m.sheets)
    nrm = sheet_normal(base[:, 0].astype(np.float64), base[:, 1].astype(np.float64), m.sheets)
    d = sigma * rng.standard_normal(n)
    coord = (base + d[:, None] * nrm).astype(np.float32)
    label = d.astype(np.float32)
    return coord, label


def init_decoder(hidden, levels, in_dim=11, seed=42, out_dim=1):
    """nn.Linear default init (uniform +-1/sqrt(fan_in)) in state_dict order, flat."""
    rng = np.random.default_rng(seed)
    out, d = [], in_dim
    for _ in range(levels):
        b = 1.0 / np.sqrt(d)
        out += [rng.uniform(-b, b, hidden * d), rng.uniform(-b, b, hidden)]
        d = hidden
    b = 1.0 / np.sqrt(d)
    out += [rng.uniform(-b, b, out_dim * d), rng.uniform(-b, b, out_dim)]
    return np.concatenate(out).astype(np.float32)
