"""Drop-in for the reference's ``utils.mapper.Mapper`` (utils/mapper.py:33).

``mapping`` (the online-training hot loop, mapper.py:600-844), ``process_frame`` (mapper.py:162-449), ``get_batch``,
``sdf`` and the pool maintenance run on the HIP kernels.  The class does NOT inherit the reference's Mapper (until round 5 it
did when the reference tree was importable, which made drop-in mode run the reference's ``determine_used_pose`` / ``init_pool``
/ ``free_pool`` while the tests and the bench ran the re-implementations below): ONE code path, the tested one, whatever is
importable.  Every public method of the reference class exists here with the reference's signature."""
from __future__ import annotations

import math
import os
import sys

import torch

import numpy as np

from ... import _lib, engine, hostcache, ops
from ... import pool as pool_mod


class _StandaloneBase:
    """Pool surface of the reference Mapper used by the hot loop (mapper.py:34-97, 452-503)."""

    def __init__(self, config, dataset, neural_points, decoders: dict):
        self.config = config
        self.silence = config.silence
        self.dataset = dataset
        self.neural_points = neural_points
        self.sdf_mlp = decoders["sdf"]
        self.sem_mlp = decoders.get("semantic")
        self.color_mlp = decoders.get("color")
        self.device = config.device
        self.dtype = config.dtype
        self.used_poses = None
        self.total_iter = 0
        self.sdf_scale = config.logistic_gaussian_ratio * config.sigma_sigmoid_m
        self.new_idx = None
        self.ba_done_flag = False
        self.adaptive_iter_offset = 0
        self.init_pool()

    def init_pool(self):
        d, f = self.device, self.dtype
        self.coord_pool = torch.empty((0, 3), device=d, dtype=f)
        self.global_coord_pool = torch.empty((0, 3), device=d, dtype=f)
        self.sdf_label_pool = torch.empty((0,), device=d, dtype=f)
        self.color_pool = torch.empty((0, self.config.color_channel), device=d, dtype=f) if self.config.color_on else None
        self.sem_label_pool = torch.empty((0,), device=d, dtype=torch.int) if getattr(self.config, "semantic_on", False) else None
        self.weight_pool = torch.empty((0,), device=d, dtype=f)
        self.time_pool = torch.empty((0,), device=d, dtype=torch.int)
        self.pool_sample_count = 0

    def free_pool(self):
        """mapper.py:589-597."""
        self.coord_pool = self.weight_pool = self.sdf_label_pool = self.time_pool = None
        self.sem_label_pool = self.color_pool = self.normal_label_pool = None

    def determine_used_pose(self):
        """mapper.py:139-160: the poses the pool samples are expressed with -- pose-graph, odometry or ground truth."""
        ds, c = self.dataset, self.config
        cur = ds.processed_frame
        if c.pgo_on:
            src = ds.pgo_poses
        elif c.track_on:
            src = ds.odom_poses
        elif getattr(ds, "gt_pose_provided", False):
            src = ds.gt_poses
        else:
            return
        # The reference uploads the whole pose history here, every frame (a synchronous copy from pageable memory: the host waits
        # for everything queued so far).  Nothing of the hot path reads it on the device, so the host keeps the snapshot and
        # `used_poses` turns it into the same tensor when somebody asks (property below).
        self._used_poses_host = np.array(src[:cur + 1], dtype=np.float64)
        self._used_poses_dev = None

    @property
    def used_poses(self):
        if self._used_poses_dev is None and self._used_poses_host is not None:
            self._used_poses_dev = torch.tensor(self._used_poses_host, device=self.device, dtype=torch.float64)
        return self._used_poses_dev

    @used_poses.setter
    def used_poses(self, value):
        self._used_poses_host, self._used_poses_dev = None, value

    def get_ba_samples(self, subsample_count):
        """mapper.py:506-524 feeds bundle_adjustment only, which is refused (see Mapper.bundle_adjustment)."""
        raise NotImplementedError("Mapper.get_ba_samples belongs to the bundle adjustment, which is outside libpinhip's hot path: "
                                  "run with ba_freq_frame: 0")

    def get_data_pool_o3d(self, down_rate=1, only_cur_data=False):
        """mapper.py:534-587 (visualisation of the sample pool, GUI / --visualize runs only): an open3d point cloud of the pool's
        global coordinates coloured by the SDF label (seismic colour map, blue = in front, red = behind)."""
        import open3d as o3d
        from matplotlib import cm
        sl = slice(-self.cur_sample_count, None, 3) if only_cur_data else slice(None, None, down_rate)
        pc = o3d.geometry.PointCloud()
        pc.points = o3d.utility.Vector3dVector(self.global_coord_pool[sl].detach().cpu().numpy().astype(np.float64))
        if self.sdf_label_pool is None:
            return pc
        lab = self.sdf_label_pool[sl].detach().cpu().numpy().astype(np.float64)
        lo = self.config.free_sample_end_dist_m * -2.0
        lab = np.clip((lab - lo) / (-2.0 * lo), 0.0, 1.0)
        pc.colors = o3d.utility.Vector3dVector(cm.get_cmap("seismic")(1.0 - lab)[:, :3].astype(np.float64))
        return pc

    def get_batch(self, global_coord=False):
        """Uniform pool sampling (mapper.py:477-503; the 'new sample' half needs process_frame)."""
        index = torch.randint(0, self.pool_sample_count, (self.config.bs,), device=self.device)
        coord = (self.global_coord_pool if global_coord else self.coord_pool)[index, :]
        color = self.color_pool[index] if self.color_pool is not None else None
        return coord, self.sdf_label_pool[index], self.time_pool[index], None, None, color, self.weight_pool[index]


class Mapper(_StandaloneBase):
    def __init__(self, config, dataset, neural_points, decoders: dict):
        super().__init__(config, dataset, neural_points, decoders)
        self._trainer = None
        # "identical inputs" mode: draw the batch indices inside get_batch, two torch.randint calls per ITERATION, exactly as the
        # reference consumes the generator (mapper.py:462-480).  Off by default: mapping() draws the indices of all its
        # iterations in two launches (iid uniform either way, but a seeded run then does not reproduce the reference's stream).
        # Set it here, as `config.draw_per_iteration`, or through PIN_DRAW_PER_ITERATION=1; INTEGRATION.md "Random numbers".
        self.draw_per_iteration = bool(getattr(config, "draw_per_iteration", False))
        self._spool = None  # device-resident sample pool (pin_slam_amd.pool.SamplePool)
        # data-parallel mapping: config.bs is the GLOBAL batch, every rank draws the same
        # batch (same seed) and trains on its contiguous shard; set by the launcher, 1 rank by default
        self.dp_rank, self.dp_world = 0, 1
        self.dp_comm = None  # pin_slam_amd.collective.RcclComm when dp_world > 1
        # how the batch is cut over the ranks (engine.MapTrainer): "spatial" = k-d boxes of the voxel grid, only the
        # halo rows cross xGMI per iteration (pin_slam_amd.dp); "dense" = contiguous index shards + an all-reduce of the
        # whole gradient table per iteration
        self.dp_mode = "spatial"
        self.static_mask = None
        self.cur_sample_count = 0
        self.cur_new_point_ratio = 0.0
        if getattr(config, "ba_freq_frame", 0) and config.ba_freq_frame > 0:
            # say it before the run starts, not at the first bundle-adjustment frame (run_replica.yaml, run_ncd_128_s.yaml)
            print(f"[pin_slam_amd] WARNING: ba_freq_frame = {config.ba_freq_frame}: Mapper.bundle_adjustment (pypose pose optimisation) is "
                  f"not part of libpinhip's hot path and will raise NotImplementedError at frame {config.ba_freq_frame}; run with "
                  f"ba_freq_frame: 0", flush=True)

    # ------------------------------------------------------------------ data pool
    _POOL_ATTRS = (("coord_pool", "coord"), ("global_coord_pool", "global_coord"), ("sdf_label_pool", "sdf_label"),
                   ("weight_pool", "weight"), ("time_pool", "ts"), ("color_pool", "color"), ("sem_label_pool", "sem_label"))

    def _pool(self) -> pool_mod.SamplePool:
        c = self.config
        C = int(c.color_channel) if c.color_on else 0
        sem = bool(getattr(c, "semantic_on", False))
        if self._spool is None or self._spool.C != C or self._spool.semantic != sem:
            self._spool = pool_mod.SamplePool(self.device, color_channels=C, semantic=sem)
            self._spool.clear()
        p = self._spool
        # somebody replaced a pool tensor (init_pool, transform_data_pool, bundle adjustment, a test): adopt it
        lab = getattr(self, "sdf_label_pool", None)
        n_ext = 0 if lab is None else lab.shape[0]
        stale = n_ext != p.n
        if not stale and p.n:
            for attr, name in self._POOL_ATTRS:
                t, v = getattr(self, attr, None), p.view(name)
                if v is not None and (t is None or t.data_ptr() != v.data_ptr()):
                    stale = True
        if stale:
            if n_ext == 0:
                p.clear()
            else:
                p.adopt(**{name: getattr(self, attr, None) for attr, name in self._POOL_ATTRS})
        return p

    def _publish_pool(self):
        """The reference's pool attributes as views of the device pool (no copies)."""
        p = self._spool
        for attr, name in self._POOL_ATTRS:
            setattr(self, attr, p.view(name))
        self.normal_label_pool = None
        self.pool_sample_count = p.n

    def dynamic_filter(self, points_torch, type_2_on: bool = True):
        """mapper.py:98-137 on the fused SDF query (SDF, analytic gradient, certainty in one launch)."""
        c, npts = self.config, self.neural_points
        q = points_torch.detach().to(torch.float32).contiguous()
        nbr, nn, _ = npts.knn(q, True)
        fs = npts.field_state(self.sdf_mlp, query_locally=True)
        sdf, grad, _, cert = ops.sdf_query(fs, q, nbr, nn, grad=type_2_on, std=False)
        static_mask = (cert < c.dynamic_certainty_thre) | (sdf < c.dynamic_sdf_ratio_thre * c.voxel_size_m)
        if type_2_on:
            static_mask &= (grad.norm(dim=-1) > c.dynamic_min_grad_norm_thre) | (cert < c.dynamic_certainty_thre)
        return static_mask

    def process_frame(self, point_cloud_torch, frame_label_torch, cur_pose_torch, frame_id: int,
                      filter_dynamic: bool = False):
        """Mapper.process_frame (mapper.py:162-449): sample the scan, grow the map, maintain the
        sample pool, find the newly observed samples.  Device-resident: the sampler writes at the
        pool tail, the filter compacts between the pool's two generations; host syncs = the
        counts the reference reads back as well."""
        c, npts = self.config, self.neural_points
        sem_on = bool(getattr(c, "semantic_on", False))
        if frame_label_torch is not None and not sem_on:
            frame_label_torch = None  # (the reference's sampler would carry them into a pool nothing reads, mapper.py:280-283)
        if self.ba_done_flag:
            raise NotImplementedError("process_frame after bundle adjustment (pool re-projection with per-frame poses)")
        if c.color_on and c.color_channel not in (1, 3):
            raise NotImplementedError("colour pool with color_channel not in (1, 3)")
        # the pose on the host: the tracker wrote this tensor from host numbers a moment ago (hostcache), so no copy back
        hostcache.stamp("pf:start")
        pose_np = np.ascontiguousarray(hostcache.to_host(cur_pose_torch), dtype=np.float64)
        origin, orientation = cur_pose_torch[:3, 3], cur_pose_torch[:3, :3]
        npts._sensor_hint = (origin, pose_np[:3, 3].copy())  # reset_local_map(origin, ...) needs it on the host as well
        scan = point_cloud_torch.detach()
        if scan.dtype != torch.float32 or not scan.is_cuda or scan.stride(1) != 1:
            scan = scan.to(device=self.device, dtype=torch.float32).contiguous()
        n_scan = scan.shape[0]
        ones = getattr(self, "_ones_mask", None)  # (a view of a cached all-true mask: no fill launch per frame; never written)
        if ones is None or ones.shape[0] < n_scan:
            ones = self._ones_mask = torch.ones(int(n_scan * 1.25) + 1024, dtype=torch.bool, device=self.device)
        self.static_mask = ones[:n_scan]
        if filter_dynamic:
            npts.reset_local_map(origin, orientation, frame_id)
            glob = torch.empty((n_scan, 3), dtype=torch.float32, device=self.device)
            ops.transform_points(scan, pose_np, glob)
            self.static_mask = self.dynamic_filter(glob)
            scan = scan[self.static_mask].contiguous()
            if frame_label_torch is not None:
                frame_label_torch = frame_label_torch[self.static_mask]
        self.dataset.static_mask = self.static_mask

        hostcache.stamp("pf:before_sampler")
        # K12: DataSampler.sample + pool append + sensor->world transform
        p = self._pool()
        n_hist = p.n
        n_new = p.append_samples(scan, pool_mod.sample_params(c, pose_np, frame_id), sem_labels=frame_label_torch if sem_on else None)
        self.cur_sample_count = n_new
        self.pool_sample_count = n_hist
        hostcache.stamp("pf:sampler_enqueued")

        # map growth (mapper.py:236-262).  The window mask of the pool filter does not depend on the map, so it
        # is queued first and its kept-count comes back in the same read-back as the surface-point count.
        tail = slice(n_hist, n_hist + n_new)
        lab_new = p.bufs[0]["sdf_label"][tail]
        filtering = (frame_id + 1) % c.pool_filter_freq == 0
        sel = cnt = None
        if c.from_sample_points and not c.from_all_samples:
            sel, cnt = ops.select_surface_points(p.bufs[0]["global_coord"][tail], lab_new,
                                                 np.float32(c.surface_sample_range_m * c.map_surface_ratio), cnt=p.counts[2:3])
        if filtering:
            p.filter_begin(pose_np[:3, 3], c.window_radius, int(c.pool_capacity))
        hostcache.stamp("pf:select_window_enqueued")
        kept = None
        if sel is not None:
            if filtering:  # the window's kept count and the surface count sit in one block: one read-back
                kept, _, n_surf = p.read_counts(3)
            else:
                n_surf, kept = int(cnt.item()), None
            update_points = sel[:n_surf]
        elif c.from_sample_points:  # from_all_samples: the reference passes sensor-frame samples here (mapper.py:238)
            update_points = p.bufs[0]["coord"][tail]
        else:
            update_points = torch.empty((scan.shape[0], 3), dtype=torch.float32, device=self.device)
            ops.transform_points(scan, pose_np, update_points)
        hostcache.stamp("pf:sync1_done")
        if c.prune_map_on and ((frame_id + 1) % c.prune_freq_frame == 0):
            if npts.prune_map(c.max_prune_certainty):
                npts.recreate_hash(None, None, True, True, frame_id)
        # (the size of the new local map is read back together with the last count of this function)
        defer = c.bs_new_sample > 0 and self.silence
        npts._defer_local_count = defer
        # (r05: the narrow brick build is queued by update() itself; PIN_DEFER_BRICKS=1 / 2 = r04's schedules, queued by
        # _process_frame_tail behind the pool filter / the certainty query -- see NeuralPoints._rebuild_bricks)
        npts._defer_bricks = os.environ.get("PIN_DEFER_BRICKS", "0") != "0"
        npts._bricks_overlap = True  # the build this update() queues runs beside the pool filter / certainty query below: narrow launches
        try:
            self.cur_new_point_ratio = npts.update(update_points, origin, orientation, frame_id)
        finally:
            npts._defer_local_count = False
            npts._defer_bricks = False
        hostcache.stamp("pf:update_done")
        try:
            self._process_frame_tail(c, npts, p, frame_id, filtering, kept, n_new, defer)
        finally:
            npts.build_pending_bricks()
            npts._bricks_overlap = False
            # an exception between update() and the read-back below must not leave the local tables at last frame's size
            if getattr(npts, "_local_count_pending", False):
                npts._finish_local_map(int(npts._cnt[2].item()))

    def _process_frame_tail(self, c, npts, p, frame_id, filtering, kept, n_new, defer):
        if not defer:
            npts.record_memory(verbose=(not self.silence))
        self.determine_used_pose()

        # K13: pool window + capacity (mapper.py:303-360); the discard draw comes after update's, as in the reference
        # (the brick build deferred by update() goes to its side stream once the filter's launches are queued: they run
        # unhindered, the build takes the device while this stream waits for the filter's counts.  Same box, 2 runs each,
        # map prep + mapping per frame: build queued by update() 2.37-2.39 ms; here 2.28-2.32; behind the certainty query
        # (PIN_DEFER_BRICKS=2) 2.38 -- then Mapper.mapping waits for it)
        late = os.environ.get("PIN_DEFER_BRICKS", "0") == "2"
        if filtering:
            self.pool_sample_count, self.cur_sample_count = p.filter_finish(int(c.pool_capacity), kept=kept,
                                                                            before_sync=None if late else npts.build_pending_bricks)
        else:
            self.cur_sample_count, self.pool_sample_count = n_new, p.n
        if not late:
            npts.build_pending_bricks()
        self._publish_pool()
        hostcache.stamp("pf:filter_done_sync3")

        # K14: newly observed close-to-surface samples (mapper.py:368-439)
        if c.bs_new_sample > 0:
            cur = self.cur_sample_count
            first = self.pool_sample_count - cur
            cert = npts._query_certainty(p.bufs[0]["global_coord"][first:first + cur], own_cell=True)
            idx, cnt = ops.new_sample_index(cert, p.bufs[0]["sdf_label"][first:first + cur], c.new_certainty_thre,
                                            np.float32(c.surface_sample_range_m * 3.0), offset=first, cnt=npts._cnt[3:4])
            hostcache.stamp("pf:certainty_enqueued")
            npts.build_pending_bricks()  # beside the count read-back below and the host work up to Mapper.mapping
            if getattr(npts, "_local_count_pending", False):
                counted, new_count = (int(v) for v in npts._cnt[2:4].tolist())  # (adjacent slots of one block: one read-back)
                npts._finish_local_map(counted)
                npts.record_memory(verbose=False)
            else:
                new_count = int(cnt.item())
            hostcache.stamp("pf:sync4_done")
            self.new_idx = idx[:new_count]
            self.adaptive_iter_offset = 0
            if c.adaptive_iters and cur > 0:
                r = new_count / cur
                if r < c.new_sample_ratio_less:
                    self.adaptive_iter_offset = -5
                elif r > c.new_sample_ratio_more:
                    self.adaptive_iter_offset = 5
                    if frame_id > c.freeze_after_frame and r > c.new_sample_ratio_restart:
                        self.adaptive_iter_offset = 10

    def bundle_adjustment(self, iter_count, window_size: int = 50, use_lie_group: bool = False):
        """mapper.py:848-937 optimises the window's poses with pypose through autograd of the SDF at re-projected pool
        samples: outside the hot path, and the reference's own code would run it on queries that are
        not differentiable here.  Refused rather than run wrongly (configurations with ba_freq_frame > 0:
        run_ncd_128_s.yaml, run_replica.yaml -- set it to 0)."""
        raise NotImplementedError("Mapper.bundle_adjustment (pypose pose optimisation) is outside libpinhip's hot path: "
                                  "run with ba_freq_frame: 0")

    def transform_data_pool(self, pose_diff_torch):
        """mapper.py:527-531: re-project the global sample coordinates after a pose-graph correction."""
        p = self._pool()
        if p.n:
            ops.transform_by_frame(p.bufs[0]["global_coord"][:p.n], p.bufs[0]["ts"][:p.n], pose_diff_torch)
        self._publish_pool()

    def get_batch(self, global_coord=False):
        """Mapper.get_batch (mapper.py:452-503): called on its own, the reference's per-call torch.randint draws
        (history, then new samples); gathers by the pool kernels.  While `mapping` runs, `self._queries_for` (a TrainBuffers) makes the same
        launch write that iteration's training queries as well."""
        c = self.config
        _queries_for = getattr(self, "_queries_for", None)
        p = self._pool()
        n = self.pool_sample_count
        new_idx = self.new_idx
        ds = self.dataset
        index_new_batch = None
        use_new = (c.bs_new_sample > 0 and new_idx is not None and not getattr(ds, "lose_track", False)
                   and not getattr(ds, "stop_status", False) and new_idx.shape[0] > 0)
        bs_new = min(new_idx.shape[0], c.bs_new_sample) if use_new else 0
        drawn = getattr(self, "_drawn", None)
        if drawn is not None and drawn["key"] == (n, c.bs - bs_new, bs_new, 0 if not use_new else new_idx.shape[0]):
            # mapping() drew the indices of all its iterations in two launches (same distribution, iid uniform)
            i = drawn["next"]
            drawn["next"] = i + 1
            index_history = drawn["hist"][i]
            index_new_batch = drawn["new"][i] if use_new else None
        elif use_new:
            index_history = torch.randint(0, n, (c.bs - bs_new,), device=self.device)
            index_new_batch = torch.randint(0, new_idx.shape[0], (bs_new,), device=self.device)
        else:
            index_history = torch.randint(0, n, (c.bs,), device=self.device)
        b = p.bufs[0]
        dev = self.device
        # data-parallel mapping: this rank gathers rows [lo, hi) of the global batch only (the index draws are
        # identical on every rank); the shard may straddle the history / new-sample boundary
        lo, hi = getattr(self, "_shard", None) or (0, c.bs)
        nb = hi - lo
        n_hist = index_history.shape[0]
        h_lo, h_hi = min(lo, n_hist), min(hi, n_hist)
        n_lo = max(lo, n_hist) - n_hist
        out = (torch.empty((nb, 3), dtype=torch.float32, device=dev), torch.empty((nb,), dtype=torch.float32, device=dev),
               torch.empty((nb,), dtype=torch.float32, device=dev), torch.empty((nb,), dtype=torch.int32, device=dev))
        color = torch.empty((nb, p.C), dtype=torch.float32, device=dev) if p.C else None
        L = _lib.lib()
        _lib.check(L.pin_gather_batch_drawn(
            (b["global_coord"] if global_coord else b["coord"]).data_ptr(), b["sdf_label"].data_ptr(), b["weight"].data_ptr(),
            b["ts"].data_ptr(), b["color"].data_ptr() if p.C else None, p.C, index_history.data_ptr() + 8 * h_lo, h_hi - h_lo,
            None if index_new_batch is None else index_new_batch.data_ptr() + 8 * n_lo,
            None if index_new_batch is None else new_idx.data_ptr(),
            nb, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(),
            None if color is None else color.data_ptr(),
            None if _queries_for is None else _queries_for.query.data_ptr(),
            0 if _queries_for is None else _queries_for.n_eik, 1 if _queries_for is None else _queries_for.dec,
            0 if _queries_for is None else _queries_for.eik_first,
            0.0 if _queries_for is None else float(np.float32(c.voxel_size_m * c.num_grad_step_ratio)),
            ops._stream()), "pin_gather_batch_drawn")
        sem = None
        if p.semantic:  # sem_label = sem_label_pool[index] (mapper.py:490-491)
            sem = torch.empty((nb,), dtype=torch.int32, device=dev)
            _lib.check(L.pin_gather_labels_drawn(
                b["sem_label"].data_ptr(), index_history.data_ptr() + 8 * h_lo, h_hi - h_lo,
                None if index_new_batch is None else index_new_batch.data_ptr() + 8 * n_lo,
                None if index_new_batch is None else new_idx.data_ptr(), nb, 1, 0, 0, sem.data_ptr(), ops._stream()),
                "pin_gather_labels_drawn")
        return out[0], out[1], out[3], None, sem, color, out[2]

    def _gather_group(self, t, it0, gn, global_coord):
        """get_batch for iterations it0 .. it0 + gn - 1 in ONE launch (pin_gather_batches_drawn): the drawn index rows of
        _draw_all, outputs [gn][bs][...] and the training queries of every iteration into t.buf.query_all."""
        c, p, drawn, buf = self.config, self._pool(), self._drawn, t.buf
        hist, new = drawn["hist"], drawn["new"]
        nb, n_hist = c.bs, hist.shape[1]
        key = (buf.group, nb, p.C, p.semantic)
        if getattr(self, "_group_key", None) != key:
            dev, G = self.device, buf.group
            self._group_out = (torch.empty((G, nb, 3), dtype=torch.float32, device=dev), torch.empty((G, nb), dtype=torch.float32, device=dev),
                               torch.empty((G, nb), dtype=torch.float32, device=dev), torch.empty((G, nb), dtype=torch.int32, device=dev),
                               torch.empty((G, nb, p.C), dtype=torch.float32, device=dev) if p.C else None,
                               torch.empty((G, nb), dtype=torch.int32, device=dev) if p.semantic else None)
            self._group_key = key
        coord, label, weight, ts, color, sem = self._group_out
        b = p.bufs[0]
        _lib.check(_lib.lib().pin_gather_batches_drawn(
            (b["global_coord"] if global_coord else b["coord"]).data_ptr(), b["sdf_label"].data_ptr(), b["weight"].data_ptr(),
            b["ts"].data_ptr(), b["color"].data_ptr() if p.C else None, p.C, hist.data_ptr() + 8 * it0 * n_hist, n_hist,
            None if new is None else new.data_ptr() + 8 * it0 * new.shape[1], None if new is None else self.new_idx.data_ptr(),
            nb, coord.data_ptr(), label.data_ptr(), weight.data_ptr(), ts.data_ptr(), None if color is None else color.data_ptr(),
            buf.query_all.data_ptr(), buf.n_eik, buf.dec, buf.eik_first, float(np.float32(c.voxel_size_m * c.num_grad_step_ratio)),
            gn, n_hist, 0 if new is None else new.shape[1], ops._stream()), "pin_gather_batches_drawn")
        if sem is not None:
            _lib.check(_lib.lib().pin_gather_labels_drawn(
                b["sem_label"].data_ptr(), hist.data_ptr() + 8 * it0 * n_hist, n_hist,
                None if new is None else new.data_ptr() + 8 * it0 * new.shape[1], None if new is None else self.new_idx.data_ptr(),
                nb, gn, n_hist, 0 if new is None else new.shape[1], sem.data_ptr(), ops._stream()), "pin_gather_labels_drawn")
        return coord, label, weight, ts, color, sem

    def _pool_records(self, t, iters):
        """Neighbour records of every pool sample, once per Mapper.mapping call, when the call draws at least
        `reuse_pool_records_ratio` x the pool (a 2^20 batch: every pool sample ~6 times per call): the neural points do not
        move while the map trains, so the records of a drawn sample are copied (pin_gather_records_drawn) and only its Eikonal
        probes are searched per iteration.  self.reuse_pool_records: None = by the ratio, True / False = forced."""
        c, n = self.config, self.pool_sample_count
        want = getattr(self, "reuse_pool_records", None)
        if want is None:
            want = iters * c.bs >= getattr(self, "reuse_pool_records_ratio", 2.0) * n
        if not want or n <= 0 or self.ba_done_flag:
            self._pool_rec = None  # (reuse is off for this call: give the table back)
            return None
        p, k, dev = self._pool(), t.fs.k, self.device
        # sized by the samples in the pool (x1.25 growth), not by the pool's capacity (1e7 samples x k x 16 B would be GBs)
        rec = getattr(self, "_pool_rec", None)
        if rec is None or rec[0].shape[0] < n or rec[0].shape[1] != k:
            cap = min(int(n * 1.25) + 1024, max(int(p.cap), n))
            rec = self._pool_rec = (torch.empty((cap, k, 4), dtype=torch.float32, device=dev),
                                    torch.empty((cap,), dtype=torch.int32, device=dev))
        ops.knn_query(t.st, p.bufs[0]["global_coord"][:n], k, out=(rec[0][:n], rec[1][:n], None), bricks=t.bricks)
        return rec

    def _records_group(self, t, it0, gn, rec):
        """knn_group from the pool's records: the samples' records copied, the probes of every iteration searched."""
        c, drawn, buf = self.config, self._drawn, t.buf
        hist, new = drawn["hist"], drawn["new"]
        n_hist, k = hist.shape[1], t.fs.k
        _lib.check(_lib.lib().pin_gather_records_drawn(
            rec[0].data_ptr(), rec[1].data_ptr(), k, hist.data_ptr() + 8 * it0 * n_hist, n_hist,
            None if new is None else new.data_ptr() + 8 * it0 * new.shape[1], None if new is None else self.new_idx.data_ptr(),
            c.bs, buf.Q, gn, n_hist, 0 if new is None else new.shape[1], buf.nbr_all.data_ptr(), buf.nn_all.data_ptr(),
            ops._stream()), "pin_gather_records_drawn")
        if buf.Q > buf.n_main:
            for j in range(gn):
                a, b = j * buf.Q + buf.n_main, (j + 1) * buf.Q
                ops.knn_query(t.st, buf.query_all[a:b], k, out=(buf.nbr_all[a:b], buf.nn_all[a:b], None), bricks=t.bricks)

    def _draws_per_iteration(self) -> bool:
        return bool(self.draw_per_iteration) or os.environ.get("PIN_DRAW_PER_ITERATION", "0") == "1"

    def _draw_all(self, iters):
        """The batch indices of `iters` get_batch calls in two torch.randint launches instead of 2 x iters (the
        reference draws per iteration, mapper.py:462-480; the draws are iid uniform either way)."""
        c, ds, new_idx, n = self.config, self.dataset, self.new_idx, self.pool_sample_count
        use_new = (c.bs_new_sample > 0 and new_idx is not None and not getattr(ds, "lose_track", False)
                   and not getattr(ds, "stop_status", False) and new_idx.shape[0] > 0)
        bs_new = min(new_idx.shape[0], c.bs_new_sample) if use_new else 0
        if n <= 0 or iters <= 0:
            return None
        if self._draws_per_iteration():
            return None  # the reference's per-iteration torch.randint calls in get_batch (a replayed / seeded random stream expects them)
        hist = torch.randint(0, n, (iters, c.bs - bs_new), device=self.device)
        new = torch.randint(0, new_idx.shape[0], (iters, bs_new), device=self.device) if use_new else None
        return dict(key=(n, c.bs - bs_new, bs_new, 0 if not use_new else new_idx.shape[0]), hist=hist, new=new, next=0)

    # ------------------------------------------------------------------ hot loop
    def _check_supported(self):
        c = self.config
        bad = []
        if getattr(c, "semantic_on", False):
            if self.sem_mlp is None: bad.append("semantic_on without a semantic decoder")
            elif not 2 <= int(self.sem_mlp.out_dim) <= 32: bad.append("semantic decoder with more than 32 heads")
            if self.dp_comm is not None: bad.append("semantic_on on the data-parallel mapper")
        if getattr(c, "color_on", False) and c.weight_i > 0 and c.color_channel != 3: bad.append("color_channel != 3")
        if c.main_loss_type != "bce": bad.append("main_loss_type != bce")
        if c.proj_correction_on or c.consistency_loss_on: bad.append("proj_correction / consistency loss")
        if c.ekional_loss_on and c.weight_e > 0 and not c.numerical_grad and not c.weighted_first and self.sdf_mlp.hidden_level != 1:
            bad.append("analytic Eikonal (numerical_grad_on False) with per-neighbour decoding and more than one decoder layer")
        if c.ekional_loss_on and c.ekional_add_to != "all": bad.append("ekional_add_to != all")
        if not c.opt_adam: bad.append("SGD")
        if c.weight_decay != 0.0: bad.append("weight_decay")
        if self.ba_done_flag: bad.append("mapping after bundle adjustment")
        if bad:
            raise NotImplementedError("Mapper.mapping on libpinhip does not cover: " + ", ".join(bad))

    def _get_trainer(self, peek_bricks: bool = False) -> engine.MapTrainer:
        c, npts = self.config, self.neural_points
        st = npts.search_state()
        fs = npts.field_state(self.sdf_mlp, query_locally=True)
        train_dec = bool(self.sdf_mlp.lout.weight.requires_grad)  # freeze_decoders (tools.py:263-292)
        eik = bool(c.ekional_loss_on and c.weight_e > 0)
        if eik and not c.numerical_grad:  # run_livox.yaml: the autograd gradient of every sample (mapper.py:677-678)
            eik = "analytic"
        t = self._trainer
        mode = None if self.dp_comm is None else ("dense" if eik == "analytic" else self.dp_mode)
        if (t is None or t.bs != c.bs or t.fs.dec.numel() != fs.dec.numel() or t.eikonal != eik or t.dp_mode != mode
                or t.fs.weighted_first != fs.weighted_first or (t.rank, t.world) != (self.dp_rank, self.dp_world)):
            t = engine.MapTrainer(st, fs, None, None, None, None, npts.local_point_ts_update, bs=c.bs,
                                  decimation=c.gradient_decimation, sigma=self.sdf_scale,
                                  weight_e=c.weight_e if eik else 0.0,
                                  eik_eps=c.voxel_size_m * c.num_grad_step_ratio, lr=c.lr, adam_eps=c.adam_eps,
                                  loss_weight_on=c.loss_weight_on, eikonal=eik, rank=self.dp_rank, world=self.dp_world,
                                  comm=self.dp_comm, dp_mode=mode or "spatial")
            self._trainer = t
        t.resize(fs)  # the local map changes size every frame: same buffers, new views
        t.st, t.ts_update, t.train_decoder = st, npts.local_point_ts_update, train_dec
        if c.color_on and c.weight_i > 0:  # colour branch (mapper.py:668-671, 802-812)
            fc = npts.field_state(self.color_mlp, query_locally=True, color=True)
            t.set_color(fc, surface_range=c.surface_sample_range_m, weight_i=c.weight_i,
                        train_decoder=bool(self.color_mlp.lout.weight.requires_grad))
        else:
            t.set_color(None)
        if getattr(c, "semantic_on", False) and c.weight_s > 0:  # semantic branch (mapper.py:664-667, 782-800)
            fsem = npts.field_state(self.sem_mlp, query_locally=True)
            t.set_semantic(fsem, heads=int(self.sem_mlp.out_dim), weight_s=c.weight_s, decimation=int(c.sem_label_decimation),
                           freespace_label_on=bool(getattr(c, "freespace_label_on", False)),
                           train_decoder=bool(self.sem_mlp.lout.weight.requires_grad))
        else:
            t.set_semantic(None)
        # (the build may still be running on its side stream: mapping() orders this stream behind it right in front of its
        # first search, after the optimiser reset, the batch draws and the first gather launch have been queued)
        b = npts._peek_bricks() if peek_bricks else npts._use_bricks()
        tf = bool(npts.temporal_local_map_on and npts.travel_dist is not None)
        t.bricks = b if (b is not None and b.mode[:2] == (tf, True) and npts.neighbor_K == b.cand_dx.shape[0]) else None
        return t

    def mapping(self, iter_count):
        """PIN map online training given fixed poses (mapper.py:600-844)."""
        self._check_supported()
        iter_count = max(1, iter_count + self.adaptive_iter_offset)
        t = self._get_trainer(peek_bricks=True)
        t.reset_optimizer(iter_count)  # a new Adam per call (mapper.py:615)
        t.defer_side_effects = False
        # (three launches per iteration instead of four: the next iteration's lazy-Adam launch sums the decoder's weight gradient
        # where its tail blocks need it -- engine.MapTrainer.step_batch; PIN_DEFER_DEC_REDUCE=0: the reduction launch, A/B runs)
        t.defer_dec_reduce = os.environ.get("PIN_DEFER_DEC_REDUCE", "1") != "0"
        if t.dp is not None:  # spatially sharded over the ranks (pin_slam_amd.dp)
            self.neural_points._use_bricks()
            self._mapping_spatial(t, iter_count)
            return
        from ...sharding import shard_range
        self._shard = shard_range(self.config.bs, self.dp_rank, self.dp_world)
        t.begin_side_effects()
        self._drawn = self._draw_all(iter_count)
        # the gather launch also writes the iteration's queries (the samples of this rank's shard + their Eikonal probes)
        fused_q = t.buf if t.buf.n_main == self._shard[1] - self._shard[0] else None
        grouped = (fused_q is not None and self.dp_world == 1 and self._drawn is not None
                   and getattr(self, "group_iterations", True))
        try:
            if grouped:
                # one GPU: the batches were all drawn above and the neural points do not move while the map trains, so one
                # gather launch and one kNN launch serve a whole group of iterations (TrainBuffers.group)
                G = t.buf.group
                outs0 = None
                want_reuse = getattr(self, "reuse_pool_records", None)
                if want_reuse is None:
                    want_reuse = iter_count * self.config.bs >= getattr(self, "reuse_pool_records_ratio", 2.0) * self.pool_sample_count
                if not want_reuse:
                    # (no per-pool search in this call: the first group's gather launch goes in front of the wait as well)
                    outs0 = self._gather_group(t, 0, min(G, iter_count), global_coord=not self.ba_done_flag)
                self.neural_points._use_bricks()  # the searches below read the cache
                reuse = self._pool_records(t, iter_count)  # large batches: one search per POOL sample and call
                t.defer_side_effects = False
                if reuse is not None:  # ... and their training-mode side effects applied once, from the draw counts
                    t.begin_deferred_side_effects(self._drawn["hist"], self._drawn["new"], self.new_idx, self.pool_sample_count)
                for it0 in range(0, iter_count, G):
                    gn = min(G, iter_count - it0)
                    outs = outs0 if (it0 == 0 and outs0 is not None) else self._gather_group(t, it0, gn, global_coord=not self.ba_done_flag)
                    if reuse is None:
                        t.knn_group(gn)
                    else:
                        self._records_group(t, it0, gn, reuse)
                    if t.fc is not None and outs[4] is None:
                        raise RuntimeError("color_on but the data pool holds no colour labels")
                    if t.defer_dec_reduce and t.can_step_group(outs[4]):
                        # the group's iterations queued by ONE foreign call (engine.MapTrainer.step_group): the plain path, and the
                        # two-stream path of the colour maps
                        t.step_group(outs[1], outs[2], outs[3], it0 + 1, gn, color_label=outs[4] if t.fc is not None else None)
                        self.total_iter += gn
                        continue
                    for j in range(gn):
                        coord, sdf_label, weight, ts, color_label, sem_label = (None if o is None else o[j] for o in outs)
                        if t.fc is not None and color_label is None:
                            raise RuntimeError("color_on but the data pool holds no colour labels")
                        t.buf.select(j)
                        t.step_batch(coord, sdf_label, weight, ts, it0 + j + 1,
                                     color_label=None if t.fc is None else
                                     (color_label if color_label.shape[1] == 3 else color_label[:, :3].contiguous()),
                                     queries_ready=True, knn_ready=True, sem_label=sem_label)
                        self.total_iter += 1
                t.buf.select(0)
                if reuse is not None:
                    t.apply_deferred_side_effects(reuse[0], reuse[1], self._pool().bufs[0]["ts"], self.pool_sample_count)
            if not grouped:
                self.neural_points._use_bricks()
            for it in range(0 if not grouped else iter_count, iter_count):
                self._queries_for = fused_q
                coord, sdf_label, ts, _, sem_label, color_label, weight = self.get_batch(global_coord=not self.ba_done_flag)
                self._queries_for = None
                if t.fc is not None and color_label is None:
                    raise RuntimeError("color_on but the data pool holds no colour labels")
                t.step_batch(coord, sdf_label, weight, ts, it + 1,
                             color_label=None if t.fc is None else
                             (color_label if color_label.shape[1] == 3 else color_label[:, :3].contiguous()),
                             queries_ready=fused_q is not None, sem_label=sem_label)
                self.total_iter += 1
        finally:
            self._queries_for = self._drawn = self._shard = None
        t.finish_optimizer()
        t.merge_side_effects()  # dp: certainty / ts side effects of the other ranks' shards, one exchange per call
        # only the rows the call's queries read have changed (features through the lazy optimiser, certainty / ts_update through
        # the queries' side effects), and the optimiser's pending words say which: the others are not copied back
        if t.lazy_finished and t.dp is None and t.comm is None and os.environ.get("PIN_ASSIGN_ALL_ROWS", "0") != "1":
            self.neural_points._changed_rows = t.lazy.state
        self.neural_points.assign_local_to_global()

    def _mapping_spatial(self, t, iter_count):
        """Mapper.mapping on a rank of the spatially sharded mapper: the batches are drawn as on one GPU (identically on every
        rank: same generator state), each rank trains on the samples of every batch that lie in its box."""
        c, p = self.config, self._pool()
        t.begin_side_effects()
        if self._draws_per_iteration():  # (it would come back as "nothing drawn": a silent no-op)
            raise NotImplementedError("draw_per_iteration / PIN_DRAW_PER_ITERATION=1 (per-iteration batch draws, as a replayed random stream expects "
                                      "them) is a one-GPU switch: the spatially sharded mapper plans its shards from the draws "
                                      "of the whole call")
        drawn = self._draw_all(iter_count)
        if drawn is not None:  # (None: an empty pool -- nothing to train on, as on one GPU)
            b = p.bufs[0]
            gc = not self.ba_done_flag
            # large batches: every rank searches the pool samples of its own box once per call (as _pool_records on one GPU)
            want = getattr(self, "reuse_pool_records", None)
            if want is None:
                want = iter_count * c.bs >= getattr(self, "reuse_pool_records_ratio", 2.0) * p.n
            self.dp_stats = t.plan_shards(b["global_coord"] if gc else b["coord"], drawn["hist"], drawn["new"], self.new_idx,
                                          num_nei_cells=c.num_nei_cells, pool_rows=p.n, pool_label=b["sdf_label"],
                                          reuse_records=bool(want) and not self.ba_done_flag)
            t.run_shards(b, gc, iter_count)
            self.total_iter += iter_count
        t.finish_optimizer()
        t.merge_side_effects()
        self.neural_points.assign_local_to_global()

    def sdf(self, x, get_std=False, min_nn_count=1, accumulate_stability=False):
        """mapper.py:940-956 (forward only)."""
        if accumulate_stability:
            raise NotImplementedError("Mapper.sdf(accumulate_stability=True) is unused by the reference")
        npts = self.neural_points
        q = x.detach().to(torch.float32).contiguous()
        nbr, nn, _ = npts.knn(q, True)
        fs = npts.field_state(self.sdf_mlp, query_locally=True)
        sdf, _, std, _ = ops.sdf_query(fs, q, nbr, nn, grad=False, certainty=False)
        return sdf, (std if (get_std and not self.config.weighted_first) else None), nn >= min_nn_count

    def get_numerical_gradient(self, x, sdf_x=None, eps=0.02, two_side=True):
        """mapper.py:986-1036 as a stand-alone call (inside `mapping` the six probes are generated by the gather launch and
        differenced in the training tile): central differences of the SDF, or one-sided ones against `sdf_x`."""
        x = x.detach().to(torch.float32)
        n = x.shape[0]
        e = torch.eye(3, dtype=torch.float32, device=x.device) * float(eps)
        if two_side:
            q = torch.cat([x + e[0], x - e[0], x + e[1], x - e[1], x + e[2], x - e[2]], 0)
            s = self.sdf(q)[0].reshape(6, n)
            return torch.stack([s[0] - s[1], s[2] - s[3], s[4] - s[5]], 1) / (2.0 * float(eps))
        q = torch.cat([x + e[0], x + e[1], x + e[2]], 0)
        s = self.sdf(q)[0].reshape(3, n)
        return (s - sdf_x.detach().reshape(1, n)).t().contiguous() / float(eps)

    def sdf_batch(self, x, bs, get_std=False, min_nn_count=1, accumulate_stability=False):
        outs = [self.sdf(x[i:i + bs], get_std, min_nn_count, accumulate_stability) for i in range(0, x.shape[0], bs)]
        sdf = torch.cat([o[0] for o in outs])
        std = torch.cat([o[1] for o in outs]) if outs and outs[0][1] is not None else None
        return sdf, std, torch.cat([o[2] for o in outs])
