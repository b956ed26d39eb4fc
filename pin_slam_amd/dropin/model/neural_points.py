"""Drop-in for the reference's ``model.neural_points.NeuralPoints`` (model/neural_points.py:29)
backed by libpinhip (HIP, gfx950).

Same constructor, attribute names and method signatures as the reference class, so that
pin_slam.py / utils.mapper / utils.tracker / utils.mesher keep working unchanged; the storage
is MI355X-first: capacity-managed structure-of-arrays buffers in HBM (no per-frame torch.cat),
an int32 hash table, a packed (xyz, ts) search mirror, and every per-frame method is a short
sequence of HIP kernel launches through the C ABI (include/pin_abi.h).  torch tensors are
memory owners and the currency of the public API only.

Differences, by design:
* tensors returned by query_feature are forward-only (no autograd graph); the fused kernels
  in Mapper / Tracker produce gradients directly.
* ``global2local`` is kept in device format (int32, PIN_NONLOCAL for non-local points); the
  property of that name converts to the reference's int64 convention (non-local -> 1,
  neural_points.py:498) for outside readers.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch
import torch.nn as nn

from ... import hostcache, _lib, ops
from ..._lib import PIN_NONLOCAL, LocalArrays, LocalParams, MapArrays, PruneParams, RehashParams, UpdateParams, check


def _p(t):
    return None if t is None else t.data_ptr()


class NeuralPoints(nn.Module):
    def __init__(self, config) -> None:
        super().__init__()
        ops.warmup()  # every code object of libpinhip loaded before the first frame needs one
        self.config = config
        self.silence = config.silence
        self.geo_feature_dim = config.feature_dim
        self.geo_feature_std = config.feature_std
        self.color_feature_dim = config.feature_dim
        self.color_feature_std = config.feature_std
        if config.feature_dim != 8:
            raise NotImplementedError("libpinhip is built for feature_dim = 8 (config.py:103)")
        self.mean_grid_sampling = False
        self.device = config.device
        self.dtype = config.dtype
        self.idx_dtype = torch.int64
        self.resolution = config.voxel_size_m
        self.buffer_size = int(config.buffer_size)
        self.temporal_local_map_on = True
        self.local_map_radius = config.local_map_radius
        self.diff_travel_dist_local = config.local_map_radius * config.local_map_travel_dist_ratio
        self.diff_ts_local = config.diff_ts_local
        self.reboot_ts = 0
        self.local_orientation = torch.eye(3, device=self.device)
        self.cur_ts = 0
        self.max_ts = 0
        self.travel_dist = None
        self.est_poses = None
        self.after_pgo = False
        self.color_on = bool(config.color_on)
        self.geo_feature_pca = self.color_feature_pca = None
        self.cur_memory_mb = 0.0
        self.memory_footprint = []

        # ---- global map: capacity-managed SoA ----
        self._n = 0
        self._cap = 0
        self._table = torch.full((self.buffer_size,), -1, dtype=torch.int32, device=self.device)
        self._alloc(1 << 16)
        # ---- local map ----
        self._m = 0
        self._lcap = 0
        self._lalloc(1 << 16)
        self.local_geo_features = nn.Parameter()
        self.local_color_features = nn.Parameter()
        self._local_mask = None
        self._g2l = None
        self._ws = None
        self._cnt = torch.zeros(4, dtype=torch.int32, device=self.device)
        self._bricks = self._brick_cache = None
        # width of the overlapped per-frame brick build (_build_bricks), read ONCE; a malformed value is an error, not "full width"
        e = os.environ.get("PIN_BRICK_BUILD_GRID")
        if e is not None and not (e.isdigit() and int(e) >= 0):
            raise ValueError(f"PIN_BRICK_BUILD_GRID must be a non-negative integer (blocks per launch, 0 = full width), got {e!r}")
        self._brick_grid_env = None if e is None else int(e)
        self._bricks_overlap = False
        self.set_search_neighborhood(num_nei_cells=config.num_nei_cells, search_alpha=config.search_alpha)

    # ------------------------------------------------------------------ storage
    def _new_arrays(self, cap):
        dev, f32, i32 = self.device, torch.float32, torch.int32
        return dict(pos=torch.empty((cap, 3), dtype=f32, device=dev), pos4=torch.empty((cap, 4), dtype=f32, device=dev),
                    orient=torch.empty((cap, 4), dtype=f32, device=dev),
                    geo=torch.zeros((cap + 1, 8), dtype=f32, device=dev),
                    color=torch.zeros((cap + 1, 8), dtype=f32, device=dev) if self.color_on else None,
                    ts_create=torch.empty((cap,), dtype=i32, device=dev), ts_update=torch.empty((cap,), dtype=i32, device=dev),
                    cert=torch.empty((cap,), dtype=f32, device=dev))

    def _alloc(self, cap):
        new = self._new_arrays(cap)
        self._spare = None
        if self._cap:
            n = self._n
            for k, t in new.items():
                if t is not None:
                    rows = n + 1 if k in ("geo", "color") else n
                    t[:rows] = self._g[k][:rows]
        self._g = new
        self._cap = cap

    def _lalloc(self, cap):
        dev, f32 = self.device, torch.float32
        self._l = dict(pos=torch.empty((cap, 3), dtype=f32, device=dev), orient=torch.empty((cap, 4), dtype=f32, device=dev),
                       geo=torch.zeros((cap + 1, 8), dtype=f32, device=dev),
                       color=torch.zeros((cap + 1, 8), dtype=f32, device=dev) if self.color_on else None,
                       cert=torch.empty((cap,), dtype=f32, device=dev),
                       ts_update=torch.empty((cap,), dtype=torch.int32, device=dev))
        self._lcap = cap

    def _workspace(self, n):
        need = _lib.lib().pin_maint_workspace_bytes(int(n))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty((int(need * 1.25),), dtype=torch.uint8, device=self.device)
        return self._ws

    def _spare_arrays(self):
        """A second set of map arrays of the current capacity: destination of the out-of-place compactions
        (prune_map, the merge of recreate_hash); the two sets swap roles when a result is adopted."""
        if getattr(self, "_spare", None) is None or self._spare["pos"].shape[0] != self._cap:
            self._spare = self._new_arrays(self._cap)
        return self._spare

    def _map_arrays(self, g=None) -> MapArrays:
        g, ma = (self._g if g is None else g), MapArrays()
        ma.table, ma.pos, ma.pos4, ma.orient = _p(self._table), _p(g["pos"]), _p(g["pos4"]), _p(g["orient"])
        ma.geo, ma.color = _p(g["geo"]), _p(g["color"])
        ma.ts_create, ma.ts_update, ma.certainty = _p(g["ts_create"]), _p(g["ts_update"]), _p(g["cert"])
        return ma

    def _local_arrays(self) -> LocalArrays:
        l, la = self._l, LocalArrays()
        la.pos, la.orient, la.geo, la.color = _p(l["pos"]), _p(l["orient"]), _p(l["geo"]), _p(l["color"])
        la.certainty, la.ts_update, la.global2local = _p(l["cert"]), _p(l["ts_update"]), _p(self._g2l)
        return la

    # ---- the reference's tensor attributes, as views of the capacity buffers ----
    neural_points = property(lambda s: s._g["pos"][:s._n])
    point_orientations = property(lambda s: s._g["orient"][:s._n])
    geo_features = property(lambda s: s._g["geo"][:s._n + 1])
    color_features = property(lambda s: s._g["color"][:s._n + 1] if s.color_on else None)
    point_ts_create = property(lambda s: s._g["ts_create"][:s._n])
    point_ts_update = property(lambda s: s._g["ts_update"][:s._n])
    point_certainties = property(lambda s: s._g["cert"][:s._n])
    local_neural_points = property(lambda s: s._l["pos"][:s._m])
    local_point_orientations = property(lambda s: s._l["orient"][:s._m])
    local_point_certainties = property(lambda s: s._l["cert"][:s._m])
    local_point_ts_update = property(lambda s: s._l["ts_update"][:s._m])
    buffer_pt_index = property(lambda s: s._table)

    @property
    def local_mask(self):
        return None if self._local_mask is None else self._local_mask[:self._n + 1].bool()

    @property
    def global2local(self):
        """Reference convention (int64; non-local -> 1, padding -> -1; neural_points.py:498-505)."""
        if self._g2l is None:
            return None
        g = self._g2l[:self._n + 1].long()
        g[g == PIN_NONLOCAL] = 1
        return g

    def is_empty(self):
        return self._n == 0

    def count(self):
        return self._n

    def local_count(self):
        return self._m

    def record_memory(self, verbose: bool = True, record_footprint: bool = True):
        point_dim = self.geo_feature_dim + 3 + 4 + (self.color_feature_dim if self.color_on else 0)
        self.cur_memory_mb = self.count() * point_dim * 4 / 1024 / 1024
        if verbose:
            print("# Global neural point: %d" % self.count())
            print("# Local  neural point: %d" % self.local_count())
            print("Current map memory consumption: {:.3f} MB".format(self.cur_memory_mb))
        if record_footprint:
            self.memory_footprint.append(self.cur_memory_mb)

    # ------------------------------------------------------------------ search state
    def set_search_neighborhood(self, num_nei_cells: int = 1, search_alpha: float = 1.0):
        dx, mv = ops.search_neighborhood(num_nei_cells, search_alpha, self.resolution)
        self.neighbor_dx = torch.from_numpy(dx.astype(np.int64)).to(self.device)
        self.neighbor_K = dx.shape[0]
        self.max_valid_dist2 = mv
        self._cand_off = torch.from_numpy(ops.candidate_offsets(dx, self.buffer_size)).to(self.device)

    def _travel(self):
        td = self.travel_dist
        if td is None:
            return None
        if td.dtype != torch.float32 or not td.is_cuda:
            td = td.to(device=self.device, dtype=torch.float32)
        return td.contiguous()

    def search_state(self, own_cell: bool = False) -> ops.SearchState:
        """Device search state; own_cell=True is the neighbourhood process_frame sets for
        query_certainty (num_nei_cells=1, search_alpha=0: the query's own cell, mapper.py:385-387)
        without touching the neighbourhood of the hot path."""
        cand, mv = self._cand_off, self.max_valid_dist2
        if own_cell:
            if getattr(self, "_cand_own", None) is None:
                dx, _ = ops.search_neighborhood(1, 0.0, self.resolution)
                self._cand_own = torch.from_numpy(ops.candidate_offsets(dx, self.buffer_size)).to(self.device)
            cand, mv = self._cand_own, 3 * (2 * self.resolution) ** 2
        return ops.SearchState(table=self._table, pos4=self._g["pos4"], cand_off=cand, n_points=self._n,
                               resolution=self.resolution, max_valid_dist2=mv,
                               travel_dist=self._travel(), cur_ts=self.cur_ts,
                               diff_travel_dist_local=self.diff_travel_dist_local, global2local=self._g2l)

    def field_state(self, decoder, query_locally=True, color=False) -> ops.FieldState:
        """FieldState over the local (or global) tables for `decoder` (a dropin Decoder)."""
        if query_locally and getattr(self, "_local_count_pending", False):
            raise RuntimeError("the local map was reset but its size has not been read back yet (Mapper.process_frame defers it)")
        l = self._l if query_locally else self._g
        if query_locally:
            feats = (self.local_color_features if color else self.local_geo_features).data
        else:
            feats = self.color_features if color else self.geo_features
        return ops.FieldState(
            feats=feats, dec=decoder.flat_params(), k=self.config.query_nn_k, hidden=decoder.hidden_dim,
            levels=decoder.hidden_level, weighted_first=self.config.weighted_first, sdf_scale=decoder.sdf_scale,
            certainty=l["cert"][:self._m if query_locally else self._n],
            orient=(l["orient"][:self._m if query_locally else self._n] if self.after_pgo else None),
            pos=l["pos"][:self._m if query_locally else self._n], out_dim=int(getattr(decoder, "out_dim", 1)))

    # ------------------------------------------------------------------ K8: update
    def update(self, points: torch.Tensor, sensor_position: torch.Tensor, sensor_orientation: torch.Tensor, cur_ts: int):
        L = _lib.lib()
        self._wait_bricks()
        points = points.to(device=self.device, dtype=torch.float32).contiguous()
        n = points.shape[0]
        ws = self._workspace(max(n, self._n + 1))
        stream = ops._stream()
        sel = torch.empty((n,), dtype=torch.int32, device=self.device)
        if self._n + n + 1 > self._cap:  # worst case: every sample is new
            self._alloc(int(max(self._cap * 1.5, self._n + n + 1)))
        up = UpdateParams()
        temporal = self.temporal_local_map_on and self.travel_dist is not None
        up.travel_dist = _p(self._travel()) if temporal else None
        up.buffer_size, up.n_points, up.capacity, up.n_max, up.cur_ts = self.buffer_size, self._n, self._cap, n, int(cur_ts)
        up.all_new = int(self.is_empty() or cur_ts == self.reboot_ts)
        up.resolution = float(np.float32(self.resolution))
        up.dist2_thre = float(np.float32(3 * self.resolution ** 2))
        up.diff_travel_dist_local = float(self.diff_travel_dist_local)
        ma = self._map_arrays()
        # the down-sampler with the one-word sort key first; a sample set too wide for it reports n_sel = -1 (nothing was
        # appended then: every kernel of pin_map_update is bounded by n_sel) and goes through the general form
        for vds in (L.pin_voxel_downsample_fast, L.pin_voxel_downsample):
            check(vds(_p(points), n, float(np.float32(self.resolution)), _p(sel), _p(self._cnt[0:1]), _p(ws), ws.numel(), stream),
                  "pin_voxel_downsample")
            check(L.pin_map_update(C.byref(ma), C.byref(up), _p(points), _p(sel), _p(self._cnt[0:1]), _p(self._cnt[1:2]),
                                   _p(ws), ws.numel(), stream), "pin_map_update")
            hostcache.stamp("up:vds_update_enqueued")
            n_sel, n_new = (int(v) for v in self._cnt[:2].tolist())  # the one host sync of update()
            hostcache.stamp("up:sync2_done")
            if n_sel >= 0:
                break
        old = self._n
        self._n = old + n_new
        # feature rows of the new points + the padding row (neural_points.py:395-411)
        g = self._g
        g["geo"][old:self._n + 1] = self.geo_feature_std * torch.randn(n_new + 1, 8, device=self.device)
        if self.color_on:
            g["color"][old:self._n + 1] = self.color_feature_std * torch.randn(n_new + 1, 8, device=self.device)
        hostcache.stamp("up:features_enqueued")
        self.reset_local_map(sensor_position, sensor_orientation, cur_ts, reboot_map=True)
        hostcache.stamp("up:local_map_bricks_enqueued")
        return n_new / max(n_sel, 1)

    # ------------------------------------------------------------------ K9: reset_local_map
    def reset_local_map(self, sensor_position: torch.Tensor, sensor_orientation: torch.Tensor, cur_ts: int,
                        use_travel_dist: bool = True, diff_ts_local: int = 50, reboot_map: bool = False):
        self._wait_bricks()
        self.cur_ts = cur_ts
        self.max_ts = max(self.max_ts, cur_ts)
        n = self._n
        if n + 1 > self._lcap:
            self._lalloc(int(max(self._lcap * 1.5, n + 1)))
        if self._g2l is None or self._g2l.numel() < n + 1:
            self._g2l = torch.empty((int((n + 1) * 1.5),), dtype=torch.int32, device=self.device)
            self._local_mask = torch.empty((int((n + 1) * 1.5),), dtype=torch.uint8, device=self.device)
        lp = LocalParams()
        # time mask (neural_points.py:442-472): travel-distance window, or a window of frames when the caller says so
        # (pin_slam.py:287 passes config.loop_local_map_by_travel_dist, False by default, for the loop-closure context)
        lp.time_mode = 0
        if self.temporal_local_map_on:
            if use_travel_dist:
                if self.travel_dist is not None:  # (pin_slam.py:273 sets it every frame before any map call)
                    lp.time_mode, lp.travel_dist = 1, _p(self._travel())
            else:
                lp.time_mode, lp.diff_ts_local = 2, int(diff_ts_local)
        lp.use_mid_ts = int(bool(self.config.use_mid_ts))
        lp.n_points, lp.cur_ts = n, int(cur_ts)
        lp.reboot_ts = int(self.reboot_ts) if reboot_map else -1
        lp.diff_travel_dist_local = float(self.diff_travel_dist_local)
        # `neural_points - sensor_position` promotes to the dtype of the position the caller hands in (float64 for
        # dataset.cur_pose_torch, float32 after a pose-graph update): the radius test runs in that type (:476-479)
        hint = getattr(self, "_sensor_hint", None)  # (Mapper.process_frame already holds this position on the host)
        if hint is not None and hint[0] is sensor_position:
            spn = np.asarray(hint[1], np.float64)
        else:
            spn = sensor_position.detach().to("cpu").to(torch.float64).numpy()
        lp.sensor_f64 = int(sensor_position.dtype == torch.float64)
        lp.sensor[0], lp.sensor[1], lp.sensor[2] = float(spn[0]), float(spn[1]), float(spn[2])
        lp.radius2 = float(self.local_map_radius ** 2)
        ws = self._workspace(n + 1)
        ma, la = self._map_arrays(), self._local_arrays()
        check(_lib.lib().pin_reset_local_map(C.byref(ma), C.byref(la), C.byref(lp), _p(self._local_mask),
                                             _p(self._cnt[2:3]), _p(ws), ws.numel(),
                                             ops._stream()), "pin_reset_local_map")
        self.local_orientation = sensor_orientation
        self._rebuild_bricks()
        if getattr(self, "_defer_local_count", False):
            # Mapper.process_frame reads the local-map size back together with its last count (one synchronisation
            # less per frame); nothing it runs in between looks at the local tables
            self._local_count_pending = True
            return
        self._finish_local_map(int(self._cnt[2].item()))

    def _finish_local_map(self, counted: int):
        """Adopt the size of the local map that pin_reset_local_map counted on the device (`counted` includes the
        padding entry) and publish the feature tables as the reference's Parameters."""
        self._local_count_pending = False
        self._m = int(counted) - 1
        self.local_geo_features = nn.Parameter(self._l["geo"][:self._m + 1])
        if self.color_on:
            self.local_color_features = nn.Parameter(self._l["color"][:self._m + 1])

    def _rebuild_bricks(self):
        """Per-frame cell-coherent cache of the hash lookups for local, time-filtered queries."""
        self._bricks = None
        self._bricks_pending = False
        if self.config.num_nei_cells > 2 or self._n == 0:
            return
        if getattr(self, "_defer_bricks", False):
            # (PIN_DEFER_BRICKS=1 / 2, the r04 schedule: Mapper.process_frame queues the build behind the pool filter /
            # behind the certainty query (build_pending_bricks).  At FULL width its two ~150-200 us launches hold every compute
            # unit, and a small launch of the caller's stream that arrives meanwhile waits behind them -- kernel trace, r04: the
            # pool filter's 25 us discard kernel took 192 us beside brick_fill.  With the narrow build below that no longer
            # happens and the build starts as soon as the local map is there.)
            self._bricks_pending = True
            return
        self._build_bricks()

    def build_pending_bricks(self):
        if getattr(self, "_bricks_pending", False):
            self._bricks_pending = False
            self._build_bricks()

    def _build_bricks(self):
        if self._brick_cache is None or self._brick_cache.cand_dx.shape[0] != self.neighbor_K:
            self._brick_cache = ops.BrickCache(self.neighbor_dx.cpu().numpy(), int(self.config.num_nei_cells), self.device)
        tf = self.temporal_local_map_on and self.travel_dist is not None
        # The build (~0.4 ms of GPU time) runs on a side stream: the rest of Mapper.process_frame (pool filter,
        # certainty query, new-sample index) is a chain of small kernels and count read-backs that leaves the GPU
        # mostly idle and does not touch the cache.  Consumers go through _use_bricks(), which orders their stream
        # behind the build.
        # (Measured and not kept, r04: a stream of the lowest queue priority -- the small launches of the caller's stream
        # still wait for compute units behind the build's 8 700-block launches, map prep 1.11 -> 1.20 ms; a stream with a
        # compute-unit mask that leaves an eighth of the device free -- hipExtStreamCreateWithCUMask makes a BLOCKING
        # stream, which serialises with torch's null stream: 1.02 -> 1.25 ms.)
        if getattr(self, "_side_stream", None) is None:
            self._side_stream = torch.cuda.Stream(device=self.device)
        side, main = self._side_stream, torch.cuda.current_stream()
        side.wait_stream(main)  # the local map it reads was written on the caller's stream
        # r05: launches at most 512 blocks wide (pin_brick_cache.build_grid).  The build is bound by its random table probes:
        # two blocks per compute unit with eight probes in flight per lane are nearly as fast as 8 700 blocks, and the small
        # launches of the caller's stream find free slots at once -- so the build is queued right here, as soon as the local
        # map exists, instead of behind the pool filter.  Same box, two runs each (frames/s, map prep ms): full width behind
        # the filter 174.4 / 0.99; 512 wide, queued here 179.3 / 0.86; 256 wide 173.9 (the mapper waits for the build);
        # 1024 wide and more, queued here: 167-171 (the small launches wait again).
        # (5.3 M points, C5: 1024 wide -- at 512 the mapper waits for the build: mapping 2.0 -> 2.13 ms)
        # Narrow only when the build really runs beside other work (Mapper.process_frame sets _bricks_overlap around its
        # update()): a reset_local_map outside it -- loop closure, pose-graph update, a tracker-only run -- is waited for
        # at once, and there the full-width launches are the faster ones.
        if getattr(self, "_bricks_overlap", False):
            g = getattr(self, "_brick_grid_env", None)  # (absent on a map un-pickled from an older run)
            self._brick_cache.build_grid = g if g is not None else (512 if self._n < 3_000_000 else 1024)
        else:
            self._brick_cache.build_grid = 0
        with torch.cuda.stream(side):
            self._bricks = self._brick_cache.build(self.search_state(), time_filtering=tf, local=True)
            self._bricks_event = side.record_event()

    def _wait_bricks(self):
        """Order the caller's stream behind a brick build that may still be reading the map arrays on the side stream
        (before anything rewrites the table / positions / global2local)."""
        self.build_pending_bricks()
        ev = getattr(self, "_bricks_event", None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def _peek_bricks(self):
        """The current brick cache (or None) WITHOUT ordering the caller's stream behind its build: for a caller that queues
        other work first and calls _use_bricks() right in front of its first search."""
        self.build_pending_bricks()
        return self._bricks

    def _use_bricks(self):
        """The current brick cache (or None), with the caller's stream ordered behind its build."""
        self.build_pending_bricks()
        ev = getattr(self, "_bricks_event", None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
            self._bricks_event = None
        return self._bricks

    # ------------------------------------------------------------------ K10
    def assign_local_to_global(self):
        """neural_points.py:515-526.  `self._changed_rows` (set by Mapper.mapping for ITS call, consumed here): int32 [>= local
        rows], non-zero for the local rows that changed since the local map was cut out -- only those go back; None: all."""
        changed_rows, self._changed_rows = getattr(self, "_changed_rows", None), None
        ma, la = self._map_arrays(), self._local_arrays()
        la.geo = _p(self.local_geo_features.data)
        if self.color_on:
            la.color = _p(self.local_color_features.data)
        if changed_rows is not None and (changed_rows.dtype != torch.int32 or changed_rows.shape[0] < self._m + 1):
            raise ValueError("changed_rows: int32, one word per local row")
        check(_lib.lib().pin_assign_local_to_global(C.byref(ma), C.byref(la), self._n, self._m,
                                                    None if changed_rows is None else changed_rows.data_ptr(), ops._stream()),
              "pin_assign_local_to_global")

    # ------------------------------------------------------------------ K1 / K2 tensor API
    def radius_neighborhood_search(self, points: torch.Tensor, time_filtering: bool = False):
        points = points.detach().to(torch.float32).contiguous()
        return ops.radius_search(self.search_state(), points, time_filtering=time_filtering)

    def knn(self, points: torch.Tensor, query_locally: bool = True, pose=None, out=None):
        """kNN record of the hot path (pin_knn_query)."""
        tf = self.temporal_local_map_on and query_locally and self.travel_dist is not None
        b = self._use_bricks()
        if b is not None and (not query_locally or b.mode[:2] != (bool(tf), True) or self.neighbor_K != b.cand_dx.shape[0]):
            b = None  # global queries / a temporarily changed neighbourhood use the direct probe
        return ops.knn_query(self.search_state(), points, self.config.query_nn_k, time_filtering=tf,
                             local=query_locally, pose=pose, out=out, bricks=b)

    def query_feature(self, query_points: torch.Tensor, query_ts: torch.Tensor = None, training_mode: bool = True,
                      query_locally: bool = True, query_geo_feature: bool = True, query_color_feature: bool = False):
        if not query_geo_feature and not query_color_feature:
            raise SystemExit("you need to at least query one kind of feature")
        if self.config.layer_norm_on or self.config.pos_encoding_band > 0:
            raise NotImplementedError("layer_norm_on / positional encoding are off in every shipped config")
        if query_locally and getattr(self, "_local_count_pending", False):
            raise RuntimeError("the local map was reset but its size has not been read back yet (Mapper.process_frame defers it)")
        q = query_points.detach().to(torch.float32).contiguous()
        nbr, nn_i32, _ = self.knn(q, query_locally)
        cert = self._l["cert"][:self._m] if query_locally else self._g["cert"][:self._n]
        tsu = self._l["ts_update"][:self._m] if query_locally else None
        k = self.config.query_nn_k

        def table(color):
            if query_locally:
                return (self.local_color_features if color else self.local_geo_features).data
            return self.color_features if color else self.geo_features

        def run(color, side_effects):
            fs = ops.FieldState(feats=table(color), dec=None, k=k, hidden=64, levels=1,
                                weighted_first=self.config.weighted_first, sdf_scale=1.0, certainty=cert,
                                orient=((self._l if query_locally else self._g)["orient"] if self.after_pgo else None),
                                pos=(self._l if query_locally else self._g)["pos"])
            ts32 = None
            if side_effects and query_ts is not None:
                ts32 = query_ts.to(torch.int32).contiguous()
            return ops.query_feature(fs, q, nbr, nn_i32, training=side_effects, certainty_rw=cert if side_effects else None,
                                     ts_update_rw=tsu if side_effects else None, query_ts=ts32)

        geo = color = None
        w = certainty = None
        if query_geo_feature:
            geo, w, certainty = run(False, training_mode)
        if query_color_feature and self.color_on:
            color, w2, c2 = run(True, training_mode and not query_geo_feature)
            if w is None:
                w, certainty = w2, c2
        return geo, color, w.unsqueeze(-1), nn_i32.long(), certainty

    def query_certainty(self, query_points: torch.Tensor):
        return self._query_certainty(query_points, own_cell=False)

    def _query_certainty(self, query_points: torch.Tensor, own_cell: bool):
        q = query_points.detach().to(torch.float32).contiguous()
        return ops.query_certainty(self.search_state(own_cell=own_cell), self._g["cert"][:max(self._n, 1)], q)

    # ------------------------------------------------------------------ post-loop maintenance (next-tier rows)
    def _rebuild_mirror(self):
        self._wait_bricks()
        ops.pack_positions(self._g["pos"], self._g["ts_create"], self._g["pos4"], 0, self._n)

    def prune_map(self, prune_certainty_thre, min_prune_count=500, global_prune=False):
        """neural_points.py:748-789 on the device (pin_prune_map: flags + ordered compaction into the spare arrays);
        the one read-back is the count the reference reads as well (`.item()`, :770)."""
        n = self._n
        if n == 0:
            return False
        self._wait_bricks()
        pp = PruneParams()
        pp.travel_dist = None if global_prune else _p(self._travel())
        pp.n_points, pp.cur_ts, pp.global_prune = n, int(self.cur_ts), int(bool(global_prune))
        pp.certainty_thre, pp.diff_travel_dist_local = float(prune_certainty_thre), float(self.diff_travel_dist_local)
        ws = self._workspace(n + 1)
        src_a = self._map_arrays()
        # count first (dst = NULL): the second set of map arrays is only spent when the result is adopted
        check(_lib.lib().pin_prune_map(C.byref(src_a), None, C.byref(pp), _p(self._cnt[3:4]), _p(ws), ws.numel(),
                                       ops._stream()), "pin_prune_map")
        n_keep = int(self._cnt[3].item())
        if n - n_keep <= min_prune_count:
            return False
        if not self.silence:
            print("# Prune neural points: ", n - n_keep)
        dst = self._spare_arrays()
        dst_a = self._map_arrays(dst)
        check(_lib.lib().pin_prune_map(C.byref(src_a), C.byref(dst_a), C.byref(pp), _p(self._cnt[3:4]), _p(ws), ws.numel(),
                                       ops._stream()), "pin_prune_map")
        self._g, self._spare = dst, None  # the compacted set becomes the map (recreate the hash next, as the reference says); the old set is released
        self._n = n_keep
        return True

    def adjust_map(self, pose_diff_torch):
        """neural_points.py:791-817: per-point SE(3) by creation frame + orientation update, one kernel."""
        self.after_pgo = True
        from ..quat import rotmat_to_quat
        self._wait_bricks()
        pd = pose_diff_torch.detach().to(device=self.device)
        dq = rotmat_to_quat(pd[:, :3, :3].to(torch.float32)).contiguous()
        used_ts = self._g["ts_create"][:self._n]
        if self.config.use_mid_ts:  # ((ts_create + ts_update) / 2).int()  (neural_points.py:803-806; once per loop closure)
            used_ts = torch.div(used_ts + self._g["ts_update"][:self._n], 2, rounding_mode="floor").to(torch.int32)
        ops.transform_by_frame(self._g["pos"][:self._n], used_ts, pd, quat=self._g["orient"][:self._n], dquat=dq)
        self._rebuild_mirror()

    def recreate_hash(self, sensor_position, sensor_orientation, kept_points: bool = True, with_ts: bool = True, cur_ts=0):
        """neural_points.py:819-908 on the device (pin_hash_rebuild): per voxel the point closest in time to `cur_ts`
        (or the most certain one) owns the table slot; kept_points=False additionally MERGES the map down to those
        points (the end of a run, pin_slam.py:521)."""
        n = self._n
        if n > 0:
            self._wait_bricks()
            rp = RehashParams()
            rp.buffer_size, rp.n_points, rp.cur_ts = self.buffer_size, n, int(cur_ts)
            rp.with_ts, rp.use_mid_ts = int(bool(with_ts)), int(bool(self.config.use_mid_ts))
            rp.resolution = float(np.float32(self.resolution))
            need = _lib.lib().pin_maint_workspace_bytes(n) + 4 * n
            if self._ws is None or self._ws.numel() < need:
                self._ws = torch.empty((int(need * 1.25),), dtype=torch.uint8, device=self.device)
            ws = self._ws
            sel = torch.empty((n,), dtype=torch.int32, device=self.device)
            src_a = self._map_arrays()
            dst = dst_a = None
            if not kept_points:
                if not self.silence:
                    print("Filter duplicated neural points")
                dst = self._spare_arrays()
                dst_a = self._map_arrays(dst)
            check(_lib.lib().pin_hash_rebuild(C.byref(src_a), None if dst_a is None else C.byref(dst_a), C.byref(rp), _p(sel),
                                              _p(self._cnt[3:4]), _p(ws), ws.numel(),
                                              ops._stream()), "pin_hash_rebuild")
            if not kept_points:
                self._n = int(self._cnt[3].item())
                self._g, self._spare = dst, None  # (the old set is released: map memory does not stay doubled)
            self._bricks = None  # the cache mirrors the table that was just rewritten
        else:
            self._table.fill_(-1)
        if sensor_position is not None:
            self.reset_local_map(sensor_position, sensor_orientation, cur_ts)
        if not kept_points:
            self.record_memory(verbose=(not self.silence))

    def clear_temp(self, clean_more: bool = False):
        """Drop everything that is rebuilt on load (neural_points.py:1035-1055) before pickling."""
        self._m = 0
        self.local_geo_features = nn.Parameter()
        self.local_color_features = nn.Parameter()
        self._local_mask = None
        self._g2l = None
        self._ws = None
        self._bricks = self._brick_cache = None
        self._side_stream = self._bricks_event = None  # (stream / event handles do not pickle)
        self._spare = None
        # shrink to size so the pickled map holds only live rows
        n = self._n
        self._g = {k: (None if t is None else t[:(n + 1 if k in ("geo", "color") else n)].clone())
                   for k, t in self._g.items()}
        self._cap = n
        self._lalloc(1)
        self._table = None

    def __setstate__(self, state):
        super().__setstate__(state)
        if self.__dict__.get("_table") is None:
            self._table = torch.full((self.buffer_size,), -1, dtype=torch.int32, device=self.device)

    def compute_feature_principle_components(self, down_rate: int = 1):
        """neural_points.py:175-179: PCA bases of the local feature tables for the GUI's feature colouring
        (visualisation only; the reference's own helper, a handful of torch ops when the decoder freezes)."""
        from utils.tools import feature_pca_torch
        _, self.geo_feature_pca = feature_pca_torch((self.local_geo_features.detach())[:-1], down_rate=down_rate,
                                                    project_data=False)
        if self.color_on:
            _, self.color_feature_pca = feature_pca_torch((self.local_color_features.detach())[:-1], down_rate=down_rate,
                                                          project_data=False)

    def get_neural_points_o3d(self, query_global: bool = True, color_mode: int = -1, random_down_ratio: int = 1):
        """neural_points.py:1073-1171, positions only: the open3d cloud pin_slam.py asks for at the end of a run (map
        bounding box, chunking for meshing, neural_points.ply).  The colour modes are GUI features: uncoloured here."""
        import open3d as o3d
        pts = (self.neural_points if query_global else self.local_neural_points)[::max(1, int(random_down_ratio))]
        pcd = o3d.geometry.PointCloud()
        pcd.points = o3d.utility.Vector3dVector(pts.detach().cpu().numpy().astype(np.float64))
        return pcd

    def get_map_o3d_bbx(self):
        """neural_points.py:1057-1070: the axis-aligned box around every neural point of the global map."""
        import open3d as o3d
        lo, hi = torch.aminmax(self.neural_points.detach(), dim=0)
        return o3d.geometry.AxisAlignedBoundingBox(lo.cpu().numpy().astype(np.float64), hi.cpu().numpy().astype(np.float64))
