"""Tensor-level wrappers over the C ABI.  torch tensors are device-memory owners only: every
function passes raw pointers + sizes to libpinhip on the current HIP stream; no torch op runs
on the hot path."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import math
import os

import numpy as np
import torch

from . import _lib
from ._lib import (PIN_GN_NSUMS, PIN_GN_REPLICAS, PIN_MAX_K, PIN_MLP_IN, PIN_NONLOCAL, Field,
                   GnParams, SearchParams, check)

PRIMES = (73856093, 19349669, 83492791)


def _ptr(t: Optional[torch.Tensor], dtype=None):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("libpinhip needs device (HIP) tensors; there is no CPU path")
    if not t.is_contiguous():
        raise RuntimeError("tensor must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"expected {dtype}, got {t.dtype}")
    return t.data_ptr()


_warmed = False


def warmup():
    """Load every code object of libpinhip now (pin_warmup): the drop-in constructors call this, so that the first frame of a
    run does not pay for the HIP runtime's lazy loading stage by stage (profiles/r03_e2e_dropin_60.json: single frames of
    88-170 ms per stage against medians below 1 ms)."""
    global _warmed
    if _warmed or not torch.cuda.is_available():
        return
    torch.cuda.current_device()  # (the runtime must be up and a device current)
    check(_lib.lib().pin_warmup(), "pin_warmup")
    _warmed = True


FP16_RANGE_MESSAGE = ("a decoder parameter is outside the range of the split-fp16 decoder image (|w| >= 65504 or non-finite, "
                      "csrc/mlp_h2.h): the tile kernels would return inf / NaN.  Set PIN_MLP=f32 to run this decoder on the fp32 image")


def status(clear: bool = False) -> int:
    """The library's sticky device status flags (pin_status; PIN_STATUS_*).  Synchronises the current stream."""
    out = C.c_int32(0)
    check(_lib.lib().pin_status(C.addressof(out), int(bool(clear)), _stream()), "pin_status")
    return int(out.value)


def raise_on_status(flags: int):
    """Turn a status word (pin_status / PIN_GN_STATE_STATUS) into an exception; clears the device word so that a caller that
    handles the error (e.g. by switching to PIN_MLP=f32 and restaging) is not stopped again by the old flag."""
    if flags & _lib.PIN_STATUS_FP16_RANGE:
        status(clear=True)
        raise RuntimeError(FP16_RANGE_MESSAGE)


def _stream():
    """Raw handle of torch's current stream ON THE CURRENT DEVICE.  (torch.cuda.current_stream() builds a Stream object
    through several Python layers, ~5 us a call -- four calls per training iteration; the two raw getters are what it ends
    in.  The device is looked up on every call: a rank that calls torch.cuda.set_device after its first op must not keep
    launching on the stream of the device it started on.)"""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def spatial_sort(points: torch.Tensor, cell: float, return_perm: bool = False):
    """points [n,3] in ascending Morton order of floor(p / cell) (pin_spatial_sort); optionally the permutation."""
    L = _lib.lib()
    n = points.shape[0]
    out = torch.empty_like(points)
    perm = torch.empty((n,), dtype=torch.int32, device=points.device) if return_perm else None
    if n == 0:
        return (out, perm) if return_perm else out
    ws = torch.empty((int(L.pin_maint_workspace_bytes(n)),), dtype=torch.uint8, device=points.device)
    check(L.pin_spatial_sort(_ptr(points, torch.float32), n, float(cell), _ptr(out), _ptr(perm), _ptr(ws), ws.numel(), _stream()),
          "pin_spatial_sort")
    return (out, perm) if return_perm else out


def search_neighborhood(num_nei_cells: int, search_alpha: float, resolution: float):
    """neighbor_dx [Kc,3] (meshgrid 'ij', x slowest) and max_valid_dist2
    (NeuralPoints.set_search_neighborhood, model/neural_points.py:910-948)."""
    n = int(num_nei_cells)
    r = np.arange(-n, n + 1, dtype=np.int64)
    g = np.stack(np.meshgrid(r, r, r, indexing="ij"), -1).reshape(-1, 3)
    dx = np.ascontiguousarray(g[(g ** 2).sum(-1) < (n + search_alpha) ** 2]).astype(np.int32)
    return dx, 3 * ((n + 1) * resolution) ** 2


def candidate_offsets(neighbor_dx: np.ndarray, buffer_size: int) -> np.ndarray:
    dx = np.ascontiguousarray(neighbor_dx, dtype=np.int32)
    out = np.empty(dx.shape[0], np.int32)
    check(_lib.lib().pin_candidate_offsets(dx.ctypes.data, dx.shape[0], int(buffer_size), out.ctypes.data),
          "pin_candidate_offsets")
    return out


def pack_positions(pos: torch.Tensor, ts_create: torch.Tensor, pos4: torch.Tensor, first: int = 0, n: int = None):
    n = pos.shape[0] - first if n is None else n
    check(_lib.lib().pin_pack_positions(_ptr(pos, torch.float32), _ptr(ts_create, torch.int32), first, n,
                                        _ptr(pos4, torch.float32), _stream()), "pin_pack_positions")
    return pos4


@dataclass
class SearchState:
    """Device-side search state of a NeuralPoints map (see pin_search_params in pin_abi.h)."""
    table: torch.Tensor            # [B] int32
    pos4: torch.Tensor             # [P,4] f32 (xyz + ts_create bits)
    cand_off: torch.Tensor         # [Kc] int32
    n_points: int
    resolution: float
    max_valid_dist2: float
    travel_dist: Optional[torch.Tensor] = None   # [n_ts] f32, None = no time filter
    cur_ts: int = 0
    diff_travel_dist_local: float = 0.0
    global2local: Optional[torch.Tensor] = None  # [P+1] int32 with PIN_NONLOCAL, None = global query

    def params(self, time_filtering=True, local=True) -> SearchParams:
        sp = SearchParams()
        sp.table = _ptr(self.table, torch.int32)
        sp.pos4 = _ptr(self.pos4, torch.float32)
        sp.cand_off = _ptr(self.cand_off, torch.int32)
        sp.travel_dist = _ptr(self.travel_dist, torch.float32) if time_filtering else None
        sp.global2local = _ptr(self.global2local, torch.int32) if local else None
        sp.buffer_size = self.table.shape[0]
        sp.n_points = int(self.n_points)
        sp.n_cand = self.cand_off.shape[0]
        sp.cur_ts = int(self.cur_ts)
        sp.diff_travel_dist_local = float(self.diff_travel_dist_local)
        sp.resolution = float(np.float32(self.resolution))
        sp.max_valid_dist2 = float(np.float32(self.max_valid_dist2))
        return sp


def radius_search(st: SearchState, query: torch.Tensor, time_filtering: bool = False):
    """NeuralPoints.radius_neighborhood_search -> (dist2 [N,Kc] f32, idx [N,Kc] int64)."""
    n, kc = query.shape[0], st.cand_off.shape[0]
    d2 = torch.empty((n, kc), dtype=torch.float32, device=query.device)
    idx = torch.empty((n, kc), dtype=torch.int64, device=query.device)
    sp = st.params(time_filtering=time_filtering, local=False)
    check(_lib.lib().pin_radius_search(C.byref(sp), _ptr(query, torch.float32), n, _ptr(d2), _ptr(idx), _stream()),
          "pin_radius_search")
    return d2, idx


class BrickCache:
    """Per-frame cell-coherent cache of the hash lookups (csrc/brick.hip).  Valid for the
    (time_filtering, local) mode it was built with and until the map / local map changes."""

    def __init__(self, neighbor_dx: np.ndarray, n_dilate: int, device="cuda"):
        if n_dilate > 2 or np.abs(neighbor_dx).max() > n_dilate:
            raise NotImplementedError("brick cache covers num_nei_cells <= 2")
        self.n_dilate = int(n_dilate)
        self.cand_dx = torch.from_numpy(np.ascontiguousarray(neighbor_dx, dtype=np.int32)).to(device)
        self.device = device
        self.counters = torch.zeros(4, dtype=torch.int32, device=device)
        self.mode = None
        self.n_bricks = self.n_entries = 0
        # the point-driven build (one table probe per point, mask-based dilation) is bit-identical to the cell-driven one but NOT
        # faster: 443 vs 405 us of kernel time per build at 2.2 M points (profiles/r04_bench_c3_kernel_stats.csv) -- off by default
        self.by_points = os.environ.get("PIN_BRICK_BUILD", "cells")[:1] == "p"
        self._host = self._event = None
        self._pending = False
        # > 0: the build's launches are at most this many blocks wide (pin_brick_cache.build_grid): for a build that runs on
        # a side stream beside other launches (NeuralPoints._build_bricks); 0 = full width, for a build the caller waits for
        self.build_grid = 0
        self._alloc(1 << 16, 1 << 18)

    def _alloc(self, max_bricks, max_entries):
        d = self.device
        dsize = 1
        while dsize < 4 * max_bricks:
            dsize *= 2
        self.dir_keys = torch.empty(dsize, dtype=torch.int64, device=d)
        self.dir_vals = torch.empty(dsize, dtype=torch.int32, device=d)
        self.dir_pack = torch.empty((dsize, 4), dtype=torch.int64, device=d)
        self.brick_keys = torch.empty(max_bricks, dtype=torch.int64, device=d)
        self.brick_mask = torch.empty(max_bricks, dtype=torch.int64, device=d)
        self.brick_base = torch.empty(max_bricks, dtype=torch.int32, device=d)
        self.entries = torch.empty((max_entries + 1, 4), dtype=torch.float32, device=d)  # + the sentinel row (pin_brick_build)
        self.max_bricks, self.max_entries, self.dsize = max_bricks, max_entries, dsize
        self.build_ws = None  # (sized per build: it depends on the number of points)

    def params(self) -> "_lib.BrickCacheC":
        bc = _lib.BrickCacheC()
        bc.dir_keys, bc.dir_vals, bc.brick_keys = self.dir_keys.data_ptr(), self.dir_vals.data_ptr(), self.brick_keys.data_ptr()
        bc.brick_mask, bc.brick_base, bc.entries = self.brick_mask.data_ptr(), self.brick_base.data_ptr(), self.entries.data_ptr()
        bc.cand_dx = self.cand_dx.data_ptr()
        bc.dir_pack = self.dir_pack.data_ptr()
        bc.dir_mask, bc.max_bricks, bc.max_entries, bc.n_dilate = self.dsize - 1, self.max_bricks, self.max_entries, self.n_dilate
        if self.build_ws is not None:
            bc.build_ws, bc.build_ws_bytes = self.build_ws.data_ptr(), self.build_ws.numel()
        bc.build_grid = int(self.build_grid)
        return bc

    def build(self, st: "SearchState", time_filtering=True, local=True, wait: bool = False):
        """(Re)build for the current map, without a host sync: the counters are read back asynchronously and
        looked at by the NEXT build, which grows the buffers if this one overflowed.  Until then nothing is
        wrong -- bricks that did not fit are published as "not cached" and their cells take the exact probe
        (identical results, slower).  wait=True checks (and rebuilds) right away."""
        self._settle()
        # size for the map at hand before the first launch (about one entry per occupied cell, a brick per ~16
        # of them); anything beyond is caught by the counters one build later
        want_b, want_e = st.n_points // 8 + 4096, st.n_points + st.n_points // 4 + 4096
        if want_b > self.max_bricks or want_e > self.max_entries:
            self._alloc(max(self.max_bricks, want_b), max(self.max_entries, want_e))
        sp = st.params(time_filtering=time_filtering, local=local)
        if self.by_points:  # scratch of the point-driven build (PIN_BRICK_BUILD=cells / by_points = False: the cell-driven one)
            need = int(_lib.lib().pin_brick_build_workspace_bytes(st.n_points, self.max_bricks))
            if self.build_ws is None or self.build_ws.numel() < need:
                self.build_ws = torch.empty((int(need * 1.25) + 1024,), dtype=torch.uint8, device=self.device)
        else:
            self.build_ws = None
        bc = self.params()
        check(_lib.lib().pin_brick_build(C.byref(sp), C.byref(bc), self.counters.data_ptr(), _stream()), "pin_brick_build")
        if self._host is None:
            self._host = torch.zeros(4, dtype=torch.int32).pin_memory()
            self._event = torch.cuda.Event()
        self._host.copy_(self.counters, non_blocking=True)
        self._event.record()
        self._pending = True
        self.mode = (bool(time_filtering), bool(local), st.n_points, st.cur_ts)
        if wait and self._settle():
            return self.build(st, time_filtering, local, wait=True)
        return self

    def _settle(self) -> bool:
        """Look at the counters of the last build; grow the buffers if it overflowed (returns True then)."""
        if not getattr(self, "_pending", False):
            return False
        self._event.synchronize()
        self._pending = False
        nb, ne, flags, _ = self._host.tolist()
        self.n_bricks, self.n_entries = nb, ne
        if flags == 0 and nb <= self.max_bricks and ne <= self.max_entries:
            return False
        self._alloc(max(self.max_bricks, int(nb * 1.5) + 1024) if (flags & 3 or nb > self.max_bricks) else self.max_bricks,
                    max(self.max_entries, int(ne * 1.5) + 1024))
        return True


def knn_query(st: SearchState, query: torch.Tensor, k: int, time_filtering=True, local=True,
              pose: Optional[np.ndarray] = None, out=None, bricks: Optional[BrickCache] = None):
    """k nearest valid candidates -> (nbr [N,k,4] f32, nn_count [N] int32, query_used [N,3]).
    ``pose`` (4x4 or 3x4, host) is applied to the query points inside the kernel."""
    n = query.shape[0]
    dev = query.device
    if out is None:
        nbr = torch.empty((n, k, 4), dtype=torch.float32, device=dev)
        nn = torch.empty((n,), dtype=torch.int32, device=dev)
        qout = torch.empty((n, 3), dtype=torch.float32, device=dev) if pose is not None else None
    else:
        nbr, nn, qout = out
    sp = st.params(time_filtering=time_filtering, local=local)
    pose_p = None
    if pose is not None:
        pose32 = np.ascontiguousarray(np.asarray(pose, dtype=np.float64)[:3, :4].astype(np.float32))
        pose_p = pose32.ctypes.data
    if bricks is not None:
        if bricks.mode is None or bricks.mode[:2] != (bool(time_filtering), bool(local)):
            raise RuntimeError("brick cache was built for another query mode")
        bc = bricks.params()
        check(_lib.lib().pin_knn_query_bricks(C.byref(sp), C.byref(bc), _ptr(query, torch.float32), n, k, pose_p,
                                              _ptr(qout), _ptr(nbr), _ptr(nn), _stream()), "pin_knn_query_bricks")
    else:
        check(_lib.lib().pin_knn_query(C.byref(sp), _ptr(query, torch.float32), n, k, pose_p, _ptr(qout), _ptr(nbr),
                                       _ptr(nn), _stream()), "pin_knn_query")
    return nbr, nn, (qout if pose is not None else query)


@dataclass
class FieldState:
    """Feature tables + decoder of the searched index space (pin_field in pin_abi.h)."""
    feats: torch.Tensor                 # [M+1, 8]
    dec: Optional[torch.Tensor]         # flat decoder parameters (None for feature-only calls)
    k: int
    hidden: int
    levels: int
    weighted_first: bool
    sdf_scale: float
    certainty: Optional[torch.Tensor] = None
    orient: Optional[torch.Tensor] = None   # only after PGO
    pos: Optional[torch.Tensor] = None      # [M,3]
    out_dim: int = 1                        # 1 = sdf head, 3 = colour heads
    dec_image: Optional[torch.Tensor] = None  # staged decoder (stage_decoder), valid for the current `dec` contents

    def stage_decoder(self):
        """Stage the decoder once for the launches that follow (pin_stage_decoder); call again after the decoder
        parameters change.  No-op for decoder shapes without a staged form."""
        nbytes = int(_lib.lib().pin_decoder_image_bytes(int(self.hidden), int(self.levels)))
        if nbytes == 0 or self.dec is None:
            self.dec_image = None
            return
        if self.dec_image is None or self.dec_image.numel() != nbytes:
            self.dec_image = torch.empty((nbytes,), dtype=torch.uint8, device=self.feats.device)
        img, self.dec_image = self.dec_image, None
        f = self.params()
        check(_lib.lib().pin_stage_decoder(C.byref(f), img.data_ptr(), nbytes, _stream()), "pin_stage_decoder")
        self.dec_image = img

    def params(self) -> Field:
        f = Field()
        f.feats = _ptr(self.feats, torch.float32)
        f.certainty = _ptr(self.certainty, torch.float32)
        f.orient = _ptr(self.orient, torch.float32)
        f.pos = _ptr(self.pos, torch.float32)
        f.dec = _ptr(self.dec, torch.float32) if self.dec is not None else None
        f.k, f.hidden, f.levels = int(self.k), int(self.hidden), int(self.levels)
        f.weighted_first = int(bool(self.weighted_first))
        f.sdf_scale = float(self.sdf_scale)
        f.out_dim = int(self.out_dim)
        if self.dec_image is not None:
            f.dec_image, f.dec_image_bytes = self.dec_image.data_ptr(), self.dec_image.numel()
        return f


def query_feature(fs: FieldState, query, nbr, nn, training=False, certainty_rw=None, ts_update_rw=None,
                  query_ts=None):
    """NeuralPoints.query_feature outputs from a kNN record: (feat, weight [N,k], certainty [N])."""
    n = query.shape[0]
    dev = query.device
    shape = (n, PIN_MLP_IN) if fs.weighted_first else (n, fs.k, PIN_MLP_IN)
    feat = torch.empty(shape, dtype=torch.float32, device=dev)
    w = torch.empty((n, fs.k), dtype=torch.float32, device=dev)
    cert = torch.empty((n,), dtype=torch.float32, device=dev)
    f = fs.params()
    check(_lib.lib().pin_query_feature(C.byref(f), _ptr(query, torch.float32), _ptr(nbr), _ptr(nn, torch.int32), n,
                                       _ptr(feat), _ptr(w), _ptr(cert), int(training), _ptr(certainty_rw),
                                       _ptr(ts_update_rw), _ptr(query_ts), _stream()), "pin_query_feature")
    return feat, w, cert


def decoder_sdf(fs: FieldState, feat: torch.Tensor):
    """Decoder.sdf on [n, 11] features."""
    n = feat.shape[0]
    out = torch.empty((n,), dtype=torch.float32, device=feat.device)
    f = fs.params()
    check(_lib.lib().pin_decoder_sdf(C.byref(f), _ptr(feat, torch.float32), n, _ptr(out), _stream()), "pin_decoder_sdf")
    return out


def sdf_query(fs: FieldState, query, nbr, nn, grad=True, std=True, certainty=True):
    """Fused interpolate + decode + analytic gradient: (sdf, grad, std, certainty)."""
    n = query.shape[0]
    dev = query.device
    sdf = torch.empty((n,), dtype=torch.float32, device=dev)
    g = torch.empty((n, 3), dtype=torch.float32, device=dev) if grad else None
    sd = torch.empty((n,), dtype=torch.float32, device=dev) if std else None
    ce = torch.empty((n,), dtype=torch.float32, device=dev) if (certainty and fs.certainty is not None) else None
    f = fs.params()
    check(_lib.lib().pin_sdf_query(C.byref(f), _ptr(query, torch.float32), _ptr(nbr), _ptr(nn, torch.int32), n,
                                   _ptr(sdf), _ptr(g), _ptr(sd), _ptr(ce), _stream()), "pin_sdf_query")
    return sdf, g, sd, ce


def color_term(fc: Optional[FieldState], colors: Optional[torch.Tensor], photometric: bool, photo_weight: float = 0.01,
               consist_weight: bool = True):
    """(ColorTerm struct, keep-alive refs) for the registration kernels, or (None, None)."""
    if fc is None or colors is None:
        return None, None
    mode = 2 if photometric else (1 if consist_weight else 0)
    if mode == 0:
        return None, None
    fc.stage_decoder()  # the colour image of the tile kernel, once per colour term (the decoder does not change under it)
    f = fc.params()
    ct = _lib.ColorTerm()
    ct.field = C.pointer(f)
    ct.colors = _ptr(colors, torch.float32)
    ct.mode, ct.photo_weight = mode, float(photo_weight)
    return ct, (f, colors)


def gn_accumulate(fs: FieldState, gp: GnParams, query, nbr, nn, sdf_labels=None, sums=None, want_points=False,
                  color=None):
    """Fused SDF + Jacobian + Gauss-Newton sums.  Returns the [64, 32] double replica buffer
    (sum over dim 0 on the host) and optionally per-point (sdf, grad)."""
    n = query.shape[0]
    dev = query.device
    if sums is None:
        sums = torch.empty((PIN_GN_REPLICAS, PIN_GN_NSUMS), dtype=torch.float64, device=dev)
    sdf = torch.empty((n,), dtype=torch.float32, device=dev) if want_points else None
    g = torch.empty((n, 3), dtype=torch.float32, device=dev) if want_points else None
    if color is not None and fs.dec_image is None:
        fs.stage_decoder()  # the tile kernel with the colour term copies both decoder images (the colour one: color_term)
    f = fs.params()
    check(_lib.lib().pin_gn_accumulate(C.byref(f), C.byref(gp), C.byref(color) if color is not None else None,
                                       _ptr(query, torch.float32), _ptr(nbr),
                                       _ptr(nn, torch.int32), _ptr(sdf_labels), n, _ptr(sums), _ptr(sdf), _ptr(g),
                                       _stream()), "pin_gn_accumulate")
    return sums, sdf, g


def solve_gn(sums: np.ndarray, lm_lambda: float):
    """Host side of implicit_reg (utils/tracker.py:656-679): normalise the weights
    (w /= 2*mean(w), tracker.py:524), LM damping, 6x6 solve in float64, expmap."""
    s = np.asarray(sums, np.float64).reshape(-1, PIN_GN_NSUMS).sum(0)
    cnt = int(round(s[29]))
    if cnt < 10:  # tracker.py:430-432
        return np.eye(4), cnt, 0.0, None
    scale = cnt / (2.0 * s[27])
    N = np.zeros((6, 6))
    iu = np.triu_indices(6)
    N[iu] = s[:21]
    N = N + N.T - np.diag(np.diag(N))
    N *= scale
    g = -scale * s[21:27]
    N_raw = N.copy()
    N = N + lm_lambda * np.diag(np.diag(N))
    t = np.linalg.solve(N, g)
    T = np.eye(4)
    ang = np.linalg.norm(t[:3])
    ax = t[:3] / ang
    S = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    T[:3, :3] = np.eye(3) + S * np.sin(ang) + (S @ S) * (1.0 - np.cos(ang))
    T[:3, 3] = t[3:]
    return T, cnt, float(s[28] / cnt * 100.0), dict(N_raw=N_raw, mse=scale * s[30] / cnt, photo_residual=float(s[31] / cnt))


# ------------------------------------------------------------------------------- training
from ._lib import TrainParams  # noqa: E402


class TrainBuffers:
    """Caller-owned scratch for one training iteration of `n_main` samples (+ Eikonal).  eikonal: True = central
    differences on every `decimation`-th sample (6 probes each), "analytic" = the autograd gradient of every sample
    (numerical_grad_on False, run_livox.yaml:27: no probes), False = off."""

    def __init__(self, n_main: int, decimation: int, k: int, hidden: int, levels: int, eikonal=True, device="cuda",
                 shard_start: int = 0, weighted_first: bool = True, group: int = 1, n_eik: Optional[int] = None):
        from .sharding import eikonal_shard
        self.n_main = int(n_main)
        self.dec = int(decimation)
        self.analytic = eikonal == "analytic"
        if self.analytic and not weighted_first and levels != 1:
            raise NotImplementedError("analytic Eikonal term (numerical_grad_on False) with per-neighbour decoding "
                                      "(weighted_first False): one-layer decoders (config/lidar_slam/run_livox.yaml)")
        self.eik_first, self.n_eik = eikonal_shard(shard_start, self.n_main, self.dec) if (eikonal and not self.analytic) else (0, 0)
        if n_eik is not None:  # capacities: the spatial shards (pin_slam_amd.dp) set the counts per iteration, set_counts()
            self.eik_first, self.n_eik, self.dec = 0, (int(n_eik) if (eikonal and not self.analytic) else 0), 1
        self.cap_main, self.cap_eik = self.n_main, self.n_eik
        self.Q = self.n_main + 6 * self.n_eik
        # `group` iterations' worth of queries / kNN records: the batches of a Mapper.mapping call are drawn up front and
        # the neural point positions do not move while the map trains, so ONE gather launch and ONE kNN launch serve
        # `group` iterations (select(j) points query / nbr / nn at iteration j of the group)
        self.group = max(1, int(group))
        self.query_all = torch.empty((self.group * self.Q, 3), dtype=torch.float32, device=device)
        self.nbr_all = torch.empty((self.group * self.Q, k, 4), dtype=torch.float32, device=device)
        self.nn_all = torch.empty((self.group * self.Q,), dtype=torch.int32, device=device)
        self._views = [(self.query_all[j * self.Q:(j + 1) * self.Q], self.nbr_all[j * self.Q:(j + 1) * self.Q],
                        self.nn_all[j * self.Q:(j + 1) * self.Q]) for j in range(self.group)]
        self.select(0)
        # (the analytic term keeps a second operand stream: sized as for twice the queries)
        nbytes = _lib.lib().pin_train_workspace_bytes(self.Q * (2 if self.analytic else 1), hidden, levels, 1 if weighted_first else k)
        self.ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=device)
        self.loss = torch.zeros((2,), dtype=torch.float64, device=device)
        self.color_ws = self.color_loss = None

    def select(self, j: int):
        self.query, self.nbr, self.nn = self._views[j]

    def set_counts(self, j: int, n_main: int, n_eik: int):
        """Shards of varying size: slot j of the (capacity-strided) group buffers holds n_main samples followed by the
        6 * n_eik probes of its Eikonal samples."""
        if n_main > self.cap_main or n_eik > self.cap_eik:
            raise ValueError("shard larger than the buffers")
        self.n_main, self.n_eik = int(n_main), int(n_eik)
        a, b = j * self.Q, j * self.Q + self.n_main + 6 * self.n_eik
        self.query, self.nbr, self.nn = self.query_all[a:b], self.nbr_all[a:b], self.nn_all[a:b]


def train_step(st: SearchState, fs: FieldState, buf: TrainBuffers, coord, sdf_label, sample_weight, sample_ts,
               certainty_rw, ts_update_rw, feat_grad, dec_grad, *, sigma, weight_e, eik_eps, loss_weight_on=False,
               global_n_main=None, global_n_eik=None, pred_out=None, bricks=None, before_forward=None,
               queries_ready=False, image_current=False, knn_ready=False, defer_weight_grad=False, defer_dec_reduce=False):
    """One Mapper.mapping iteration up to (not including) the optimiser step: queries -> kNN
    -> fused forward/loss/backward.  Gradients accumulate into feat_grad / dec_grad.
    `before_forward()` runs between the kNN and the forward pass (the lazy optimiser's catch-up).
    image_current: fs.dec_image holds the decoder's current parameters (LazyAdam keeps it so when it is handed the
    image) -- the launch sequence then has no staging kernel.
    defer_weight_grad: stop after the tile kernel; train_weight_grad(buf, dec_grad) finishes the step (the decoder's
    weight gradient and the loss sums), possibly on another stream.
    defer_dec_reduce: the decoder's weight gradient stays where the weight-gradient launch wrote it (no reduction launch, no
    loss sums); train_deferred_partial() says where -- hand that to the optimiser's decoder step (LazyAdam, dense[8])."""
    L = _lib.lib()
    s = _stream()
    if not queries_ready:  # (pin_gather_batch_drawn can write them in its own launch)
        check(L.pin_train_make_queries(_ptr(coord, torch.float32), buf.n_main, buf.n_eik, buf.dec, buf.eik_first,
                                       float(np.float32(eik_eps)),
                                       _ptr(buf.query), s), "pin_train_make_queries")
    if not knn_ready:  # (a group of iterations can be searched in one launch, TrainBuffers.group)
        knn_query(st, buf.query, fs.k, out=(buf.nbr, buf.nn, None), bricks=bricks)
    if before_forward is not None:
        before_forward()
    tp = TrainParams()
    tp.n_main, tp.n_eik, tp.loss_weight_on = buf.n_main, buf.n_eik, int(bool(loss_weight_on))
    tp.sigma, tp.weight_e, tp.eik_eps = float(sigma), float(weight_e), float(np.float32(eik_eps))
    tp.inv_n_main = 1.0 / float(global_n_main or buf.n_main)
    tp.inv_n_eik = 1.0 / float(global_n_eik or max(buf.n_eik, 1))
    tp.eik_analytic = int(buf.analytic)
    tp.dec_image_current = int(bool(image_current) and fs.dec_image is not None)
    if buf.analytic:  # mean over every sample of the (global) batch, mapper.py:778-781
        tp.inv_n_eik = tp.inv_n_main
    tp.defer_weight_grad = int(bool(defer_weight_grad))
    tp.defer_dec_reduce = int(bool(defer_dec_reduce) and dec_grad is not None)
    f = fs.params()
    buf.last_call = (f, tp)
    check(L.pin_train_step(C.byref(f), C.byref(tp), _ptr(buf.query), _ptr(buf.nbr), _ptr(buf.nn),
                           _ptr(sdf_label, torch.float32), _ptr(sample_weight), _ptr(sample_ts), _ptr(certainty_rw),
                           _ptr(ts_update_rw), _ptr(feat_grad, torch.float32), _ptr(dec_grad), _ptr(buf.loss),
                           _ptr(pred_out), _ptr(buf.ws), buf.ws.numel() * 4, s), "pin_train_step")
    return buf.loss


def train_deferred_partial():
    """(address, slots, n, scale) of the decoder gradient the last train_step(..., defer_dec_reduce=True) of this thread left
    as slot copies in its workspace, or None if that call reduced it as usual (a path without the slot copies)."""
    ptr, slots, n, scale = C.c_void_p(), C.c_int32(), C.c_int64(), C.c_float()
    rc = _lib.lib().pin_train_deferred_partial(C.addressof(ptr), C.addressof(slots), C.addressof(n), C.addressof(scale))
    if rc != 0 or not ptr.value:
        return None
    return (ptr.value, slots.value, n.value, scale.value)


def train_weight_grad(buf: TrainBuffers, dec_grad):
    """Second half of a train_step(..., defer_weight_grad=True) on `buf`: pin_train_weight_grad on the current stream."""
    f, tp = buf.last_call
    check(_lib.lib().pin_train_weight_grad(C.byref(f), C.byref(tp), _ptr(dec_grad), _ptr(buf.loss), _ptr(buf.ws), buf.ws.numel() * 4,
                                           _stream()), "pin_train_weight_grad")


def color_workspace(buf: TrainBuffers, fc: FieldState):
    """The colour term's scratch on `buf` (allocated with its first use)."""
    if buf.color_ws is None:
        nbytes = _lib.lib().pin_train_workspace_bytes(buf.cap_main, fc.hidden, fc.levels, 1 if fc.weighted_first else fc.k) + 256  # (capacity: shards vary)
        buf.color_ws = torch.empty((nbytes // 4 + 1,), dtype=torch.float32, device=buf.query.device)
        buf.color_loss = torch.zeros((1,), dtype=torch.float64, device=buf.query.device)


def train_color_step(fc: FieldState, buf: TrainBuffers, sdf_label, color_label, sample_weight, feat_grad, dec_grad, *,
                     surface_range, weight_i=1.0, loss_weight_on=False, image_current=False, surface_count=None,
                     global_n_main=None):
    """Colour term of a training iteration; call after train_step (reuses its queries / kNN)."""
    color_workspace(buf, fc)
    tp = _lib.TrainColorParams()
    tp.n_main, tp.loss_weight_on = buf.n_main, int(bool(loss_weight_on))
    tp.surface_range, tp.weight_i = float(surface_range), float(weight_i)
    tp.dec_image_current = int(bool(image_current) and fc.dec_image is not None)
    if surface_count is not None:  # (a shard of a larger batch: device int32, the batch's number of surface samples)
        tp.surface_count = surface_count.data_ptr()
        tp.n_main_global = int(global_n_main or 0)
    f = fc.params()
    check(_lib.lib().pin_train_color_step(C.byref(f), C.byref(tp), _ptr(buf.query), _ptr(buf.nbr), _ptr(buf.nn),
                                          _ptr(sdf_label, torch.float32), _ptr(color_label, torch.float32),
                                          _ptr(sample_weight), _ptr(feat_grad, torch.float32), _ptr(dec_grad),
                                          _ptr(buf.color_loss), _ptr(buf.color_ws), buf.color_ws.numel() * 4, _stream()),
          "pin_train_color_step")
    return buf.color_loss


# ------------------------------------------------------------------------------- semantic head (csrc/sem.h)
def sem_select(labels: torch.Tensor, freespace_label_on: bool, decimation: int, out=None):
    """The samples the semantic loss runs over (mapper.py:786-799): (selected uint8 [n], count int32 [1])."""
    n = labels.shape[0]
    sel, cnt = out if out is not None else (torch.empty((n,), dtype=torch.uint8, device=labels.device),
                                            torch.empty((1,), dtype=torch.int32, device=labels.device))
    check(_lib.lib().pin_sem_select(_ptr(labels, torch.int32), n, int(bool(freespace_label_on)), max(1, int(decimation)), _ptr(sel), _ptr(cnt),
                                    _stream()), "pin_sem_select")
    return sel, cnt


def train_sem_step(fsem: FieldState, buf: "TrainBuffers", labels, selected, count, feat_grad, dec_grad, *, heads: int, weight_s: float):
    """Semantic term of a training iteration; call after train_step (reuses the queries / kNN records of its main samples).
    fsem: geometry feature table + the semantic decoder's flat parameters.  Returns the device double the kernel adds
    sum(-sem_pred[label]) over the selected samples to (the reference's loss = that / count)."""
    if buf.n_main == 0:
        return getattr(buf, "sem_loss", None)
    nbytes = _lib.lib().pin_sem_workspace_bytes(buf.cap_main, fsem.hidden, fsem.levels, 1 if fsem.weighted_first else fsem.k)
    if getattr(buf, "sem_ws", None) is None or buf.sem_ws.numel() * 4 < nbytes:
        buf.sem_ws = torch.empty((nbytes // 4 + 1,), dtype=torch.float32, device=buf.query.device)
    if getattr(buf, "sem_loss", None) is None:
        buf.sem_loss = torch.zeros((1,), dtype=torch.float64, device=buf.query.device)
    sp = _lib.SemParams()
    sp.n_main, sp.heads, sp.weight_s = buf.n_main, int(heads), float(weight_s)
    sp.labels, sp.selected, sp.count = _ptr(labels, torch.int32), _ptr(selected, torch.uint8), _ptr(count, torch.int32)
    f = fsem.params()
    check(_lib.lib().pin_train_sem_step(C.byref(f), C.byref(sp), _ptr(buf.query), _ptr(buf.nbr), _ptr(buf.nn), _ptr(feat_grad, torch.float32),
                                        _ptr(dec_grad), _ptr(buf.sem_loss), _ptr(buf.sem_ws), buf.sem_ws.numel() * 4, _stream()),
          "pin_train_sem_step")
    return buf.sem_loss


def sem_query(fsem: FieldState, query, nbr, nn, heads: int, want_labels=True, want_logprob=False):
    """Semantic prediction at query points: (labels int32 [n] = argmax of the (weighted) log-probabilities, logprob [n, heads])."""
    n = query.shape[0]
    dev = query.device
    lab = torch.empty((n,), dtype=torch.int32, device=dev) if want_labels else None
    lp = torch.empty((n, int(heads)), dtype=torch.float32, device=dev) if want_logprob else None
    f = fsem.params()
    check(_lib.lib().pin_sem_query(C.byref(f), _ptr(query, torch.float32), _ptr(nbr), _ptr(nn, torch.int32), n, int(heads), _ptr(lab), _ptr(lp),
                                   _stream()), "pin_sem_query")
    return lab, lp


def decoder_sem(fsem: FieldState, feat: torch.Tensor, heads: int, raw: bool = False):
    """Decoder.sem_label_prob (raw: Decoder.mlp) on [n, 11] decoder inputs -> [n, heads]."""
    n = feat.shape[0]
    out = torch.empty((n, int(heads)), dtype=torch.float32, device=feat.device)
    f = fsem.params()
    check(_lib.lib().pin_decoder_sem(C.byref(f), _ptr(feat, torch.float32), n, int(heads), int(bool(raw)), _ptr(out), _stream()), "pin_decoder_sem")
    return out


def adam_step(param, grad, exp_avg, exp_avg_sq, step, lr=0.01, beta1=0.9, beta2=0.99, eps=1e-15, zero_grad=True):
    check(_lib.lib().pin_adam_step(_ptr(param, torch.float32), _ptr(grad, torch.float32), _ptr(exp_avg, torch.float32),
                                   _ptr(exp_avg_sq, torch.float32), param.numel(), int(step), float(lr), float(beta1),
                                   float(beta2), float(eps), int(bool(zero_grad)), _stream()), "pin_adam_step")


def mark_rows(nbr: torch.Tensor, row_flags: torch.Tensor):
    """row_flags[idx] = 1 for every valid neighbour of the kNN records `nbr` [Q, k, 4]."""
    check(_lib.lib().pin_mark_rows(_ptr(nbr, torch.float32), nbr.numel() // 4, _ptr(row_flags, torch.uint8), _stream()),
          "pin_mark_rows")


def adam_step_rows(param, grad, exp_avg, exp_avg_sq, row_flags, step, lr=0.01, beta1=0.9, beta2=0.99, eps=1e-15,
                   zero_grad=True):
    """Adam over the rows flagged in `row_flags` only (exact when the unflagged rows have g = m = v = 0)."""
    rows, width = param.shape[0], param.numel() // param.shape[0]
    check(_lib.lib().pin_adam_step_rows(_ptr(param, torch.float32), _ptr(grad, torch.float32), _ptr(exp_avg, torch.float32),
                                        _ptr(exp_avg_sq, torch.float32), rows, width, _ptr(row_flags, torch.uint8), int(step),
                                        float(lr), float(beta1), float(beta2), float(eps), int(bool(zero_grad)), _stream()),
          "pin_adam_step_rows")


_ADAM_COEF = {}


def adam_coef(lr, t_max, beta1=0.9, beta2=0.99, device="cuda"):
    """[2][t_max+1] table of lr/(1-beta1^t) and 1/sqrt(1-beta2^t), rounded exactly as pin_adam_step rounds them."""
    key = (float(lr), float(beta1), float(beta2), str(device))
    tab = _ADAM_COEF.get(key)
    if tab is None or tab[1] < t_max:
        cap = max(64, int(t_max) * 2)
        lr32, b1, b2 = float(np.float32(lr)), float(np.float32(beta1)), float(np.float32(beta2))
        a = np.zeros((2, cap + 1), dtype=np.float32)
        for t in range(1, cap + 1):
            a[0, t] = np.float32(lr32 / (1.0 - math.pow(b1, float(t))))
            a[1, t] = np.float32(1.0 / math.sqrt(1.0 - math.pow(b2, float(t))))
        tab = (torch.from_numpy(a).to(device), cap)
        _ADAM_COEF[key] = tab
    return tab


class LazyAdam:
    """Exact Adam over an 8-wide feature table that only visits the rows an iteration reads (pin_adam_lazy_*):
    bit-identical to adam_step over the whole table every iteration.  reset() per Mapper.mapping call,
    prepare(records, t) before the forward pass of iteration t (a row settles the step it still owes from the iteration
    that last read it, then the gradient-free steps in between), flush() at the end."""

    def __init__(self, lr=0.01, beta1=0.9, beta2=0.99, eps=1e-15):
        self.lr, self.b1, self.b2, self.eps = float(lr), float(beta1), float(beta2), float(eps)
        self.state = self.flags = None
        self.t = 0
        # batches with at least this many records per table row go through the row-parallel form (pin_adam_lazy_prepare_rows).
        # Measured: 0.09 records per row (C3: 210 k records, 2.2 M rows) 1.26 vs 1.40 ms of mapping in favour of the records;
        # 0.76 (a rank of 8 at C4) 0.610 vs 0.577 ms per iteration and 6 (C4 on one GPU) 4.29 vs 3.89 ms in favour of the rows
        self.rows_form_ratio = float(os.environ.get("PIN_LAZY_ROWS_RATIO", "0.5"))

    def reset(self, rows, t_max, device):
        if self.state is None or self.state.shape[0] < rows:
            self.state = torch.zeros((int(rows * 1.25) + 1024,), dtype=torch.int32, device=device)  # pending step per row
            self.flags = torch.zeros((int(rows * 1.25) + 1024,), dtype=torch.uint8, device=device)  # (kept clear by the kernel)
        else:
            self.state.zero_()
        self.coef, self.t_max = adam_coef(self.lr, t_max, self.b1, self.b2, device)
        self.t = 0
        self.rows_launches = 0  # prepare() calls of this lifetime that took the row-parallel form (tests read it)

    @staticmethod
    def _dense(dense):
        if dense is None:
            return None
        d = _lib.AdamDense()
        d.param, d.grad, d.exp_avg, d.exp_avg_sq = (_ptr(t, torch.float32) for t in dense[:4])
        d.n = dense[0].numel()
        if len(dense) > 4 and dense[4] is not None:  # (image, hidden, levels, out_dim): the staged decoder follows the step
            d.image, d.hidden, d.levels, d.out_dim = dense[4].data_ptr(), int(dense[5]), int(dense[6]), int(dense[7])
        if len(dense) > 8 and dense[8] is not None:  # (address, slots, n, scale) of train_deferred_partial(): the step's weight gradient
            ptr, slots, n, scale = dense[8]
            if n != d.n:
                raise ValueError("the deferred weight gradient belongs to a decoder of another size")
            d.grad_partial, d.partial_slots, d.partial_scale = ptr, int(slots), float(scale)
        return d

    def prepare(self, nbr, param, grad, m, v, step, dense=None):
        """`dense` = (param, grad, exp_avg, exp_avg_sq[, image, hidden, levels, out_dim]) of a dense tensor (the decoder):
        its step `step - 1` rides along in the same launch (identical to adam_step(..., step - 1, lr) with zero_grad);
        nothing at step 1.  With the decoder's staged image (FieldState.stage_decoder) every updated parameter is written
        through to it."""
        if step > self.t_max:
            raise ValueError("more iterations than reset() was sized for")
        if step <= self.t:
            raise ValueError("steps must grow within one optimiser lifetime (the step elects one owner per row and call)")
        d = self._dense(dense)
        n_rec = nbr.numel() // 4
        if n_rec >= self.rows_form_ratio * param.shape[0]:  # many visits per row: flag the rows, settle them in one pass over the table
            check(_lib.lib().pin_adam_lazy_prepare_rows(_ptr(nbr, torch.float32), n_rec, _ptr(param, torch.float32), _ptr(grad, torch.float32),
                                                        _ptr(m, torch.float32), _ptr(v, torch.float32), self.state.data_ptr(),
                                                        self.flags.data_ptr(), param.shape[0], int(step), _ptr(self.coef), self.t_max,
                                                        self.b1, self.b2, self.eps, C.byref(d) if d is not None else None, _stream()),
                  "pin_adam_lazy_prepare_rows")
            self.t = int(step)
            self.rows_launches += 1
            return
        check(_lib.lib().pin_adam_lazy_prepare(_ptr(nbr, torch.float32), nbr.numel() // 4, _ptr(param, torch.float32),
                                               _ptr(grad, torch.float32), _ptr(m, torch.float32), _ptr(v, torch.float32),
                                               self.state.data_ptr(), int(step),
                                               _ptr(self.coef), self.t_max, self.b1, self.b2, self.eps,
                                               C.byref(d) if d is not None else None, _stream()), "pin_adam_lazy_prepare")
        self.t = int(step)

    def step_dense(self, dense, step):
        """The dense rider's step `step` alone (what prepare(step + 1) would apply), as its own small launch: lets a
        caller run the decoder's step on another stream, behind that stream's weight gradient."""
        d = self._dense(dense)
        check(_lib.lib().pin_adam_lazy_flush(None, None, None, None, None, 0, int(step), _ptr(self.coef), self.t_max, self.b1,
                                             self.b2, self.eps, C.byref(d), _stream()), "pin_adam_lazy_flush")

    def flush(self, param, grad, m, v, dense=None):
        """Settle every touched row (and the dense tensor's last step) at the step of the last prepare()."""
        if self.t == 0:
            return
        d = self._dense(dense)
        check(_lib.lib().pin_adam_lazy_flush(_ptr(param, torch.float32), _ptr(grad, torch.float32), _ptr(m, torch.float32),
                                             _ptr(v, torch.float32), self.state.data_ptr(), param.shape[0], self.t,
                                             _ptr(self.coef), self.t_max, self.b1, self.b2, self.eps,
                                             C.byref(d) if d is not None else None, _stream()), "pin_adam_lazy_flush")
        self.t = 0


def gather_batch(pool_coord, pool_label, pool_weight, pool_ts, index, out):
    """Mapper.get_batch gathers; `out` = (coord [n,3], label [n], weight [n] | None, ts [n] | None)."""
    coord, label, weight, ts = out
    check(_lib.lib().pin_gather_batch(_ptr(pool_coord, torch.float32), _ptr(pool_label, torch.float32),
                                      _ptr(pool_weight), _ptr(pool_ts), _ptr(index, torch.int32), index.numel(),
                                      _ptr(coord), _ptr(label), _ptr(weight), _ptr(ts), _stream()), "pin_gather_batch")
    return out


INTENSITY = (0.299, 0.587, 0.114)  # color_to_intensity, utils/tools.py:408-410


def decoder_color(fs: FieldState, feat: torch.Tensor):
    """Decoder.regress_color on [n, 11] features -> [n, 3]."""
    n = feat.shape[0]
    out = torch.empty((n, 3), dtype=torch.float32, device=feat.device)
    f = fs.params()
    check(_lib.lib().pin_decoder_color(C.byref(f), _ptr(feat, torch.float32), n, _ptr(out), _stream()), "pin_decoder_color")
    return out


def color_query(fc: FieldState, query, nbr, nn, kappa=INTENSITY, want_color=True, want_grad=True):
    """Colour prediction [n,3], value = kappa . colour [n] and its gradient [n,3]."""
    n = query.shape[0]
    dev = query.device
    col = torch.empty((n, 3), dtype=torch.float32, device=dev) if want_color else None
    val = torch.empty((n,), dtype=torch.float32, device=dev)
    g = torch.empty((n, 3), dtype=torch.float32, device=dev) if want_grad else None
    kap = np.ascontiguousarray(np.asarray(kappa, dtype=np.float32))
    f = fc.params()
    check(_lib.lib().pin_color_query(C.byref(f), _ptr(query, torch.float32), _ptr(nbr), _ptr(nn, torch.int32), n,
                                     kap.ctypes.data, _ptr(col), _ptr(val), _ptr(g), _stream()), "pin_color_query")
    return col, val, g


# ------------------------------------------------------------------------------- process_frame data path
def query_certainty(st: SearchState, certainty: torch.Tensor, points: torch.Tensor, out: Optional[torch.Tensor] = None):
    """NeuralPoints.query_certainty over the global map with the neighbourhood held by `st`."""
    n = points.shape[0]
    if out is None:
        out = torch.empty((n,), dtype=torch.float32, device=points.device)
    sp = st.params(time_filtering=False, local=False)
    check(_lib.lib().pin_query_certainty(C.byref(sp), _ptr(certainty, torch.float32), _ptr(points, torch.float32), n,
                                         _ptr(out, torch.float32), _stream()), "pin_query_certainty")
    return out


def pool_workspace(n: int, device, extra: int = 0) -> torch.Tensor:
    return torch.empty((_lib.lib().pin_pool_workspace_bytes(n) + extra,), dtype=torch.uint8, device=device)


def new_sample_index(certainty, sdf_label, certainty_thre, label_thre, offset=0, ws=None, cnt=None):
    """where(certainty < thre & |label| < label_thre) + offset -> (int64 index buffer [n], count tensor [1]).
    cnt: where the count goes (an int32 [1] view of a caller's block of counts: one read-back for several counts)."""
    n = certainty.shape[0]
    dev = certainty.device
    idx = torch.empty((n,), dtype=torch.int64, device=dev)
    if cnt is None:
        cnt = torch.empty((1,), dtype=torch.int32, device=dev)
    if ws is None or ws.numel() < _lib.lib().pin_pool_workspace_bytes(n) + n:
        ws = pool_workspace(n, dev, extra=n)
    check(_lib.lib().pin_new_sample_index(_ptr(certainty, torch.float32), _ptr(sdf_label, torch.float32), n,
                                          float(certainty_thre), float(label_thre), int(offset), _ptr(idx), _ptr(cnt),
                                          _ptr(ws), ws.numel(), _stream()), "pin_new_sample_index")
    return idx, cnt


def select_surface_points(rows, sdf_label, label_thre, ws=None, cnt=None):
    """rows[|sdf_label| < label_thre] -> (buffer [n,3], count tensor [1]); order preserved.  cnt: as in new_sample_index."""
    n = sdf_label.shape[0]
    dev = rows.device
    out = torch.empty((n, 3), dtype=torch.float32, device=dev)
    if cnt is None:
        cnt = torch.empty((1,), dtype=torch.int32, device=dev)
    if ws is None or ws.numel() < _lib.lib().pin_pool_workspace_bytes(n) + n:
        ws = pool_workspace(n, dev, extra=n)
    check(_lib.lib().pin_select_surface_points(_ptr(rows, torch.float32), _ptr(sdf_label, torch.float32), n, float(label_thre),
                                               _ptr(out), _ptr(cnt), _ptr(ws), ws.numel(), _stream()),
          "pin_select_surface_points")
    return out, cnt


def transform_points(points: torch.Tensor, pose, out: Optional[torch.Tensor] = None):
    """transform_torch: points [N, >=3] (unit column stride) -> out [N,3] = R p + t (float32)."""
    if not (points.is_cuda and points.dtype == torch.float32 and points.stride(1) == 1):
        raise RuntimeError("points must be a float32 device tensor with unit column stride")
    n = points.shape[0]
    if out is None:
        out = torch.empty((n, 3), dtype=torch.float32, device=points.device)
    T = np.ascontiguousarray(np.asarray(pose, np.float64)[:3, :4])
    check(_lib.lib().pin_transform_points(points.data_ptr(), points.stride(0), n, T.ctypes.data, _ptr(out, torch.float32),
                                          _stream()), "pin_transform_points")
    return out


def transform_by_frame(points: torch.Tensor, frame: torch.Tensor, pose: torch.Tensor, quat: Optional[torch.Tensor] = None,
                       dquat: Optional[torch.Tensor] = None):
    """In place: points[i] <- pose[frame[i]] applied (float32), optionally quat[i] <- dquat[frame[i]] * quat[i]."""
    T = pose.detach().to(device=points.device, dtype=torch.float32)[:, :3, :4].contiguous()
    check(_lib.lib().pin_transform_by_frame(_ptr(points, torch.float32), points.shape[0], _ptr(frame, torch.int32), _ptr(T),
                                            T.shape[0], _ptr(quat), _ptr(dquat), _stream()), "pin_transform_by_frame")
    return points


def gather_rows(src: torch.Tensor, index: torch.Tensor, out: Optional[torch.Tensor] = None):
    """out[i] = src[index[i]] for a [n, width] float32 pool (color_pool gather of Mapper.get_batch)."""
    width = src.shape[1]
    if out is None:
        out = torch.empty((index.numel(), width), dtype=torch.float32, device=src.device)
    check(_lib.lib().pin_gather_rows(_ptr(src, torch.float32), width, _ptr(index, torch.int32), index.numel(), _ptr(out),
                                     _stream()), "pin_gather_rows")
    return out
