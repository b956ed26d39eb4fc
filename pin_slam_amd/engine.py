"""Host-side drivers of the two hot loops, on top of the C ABI (pin_slam_amd.ops):

* :class:`GNTracker`  -- Tracker.tracking / registration_step (utils/tracker.py:43-225, 367-611):
  `track` is device-resident: per Gauss-Newton iteration one kNN launch (pose read from the device state), one
  fused SDF+Jacobian+normal-equation launch and a one-wave 6x6 solve, ONE 512-byte read-back per call; `step`
  (registration_step) is host-driven: one read-back of the sums and a float64 solve on the host per call.
* :class:`MapTrainer` -- Mapper.mapping (utils/mapper.py:600-844): per iteration batch gather,
  query generation, kNN, fused forward/loss/backward, (world > 1: RCCL all-reduce through the C ABI), Adam.

The drop-in classes in ``pin_slam_amd.dropin`` wrap these with the reference's signatures.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np
import torch

from . import ops
from . import _lib
from ._lib import PIN_GN_NSUMS, PIN_GN_REPLICAS, GnParams, check


class GNTracker:
    def __init__(self, st: ops.SearchState, fs: ops.FieldState, gp: GnParams, lm_lambda: float, n_max: int):
        self.st, self.fs, self.gp, self.lm_lambda = st, fs, gp, lm_lambda
        dev = fs.feats.device
        self.nbr = torch.empty((n_max, fs.k, 4), dtype=torch.float32, device=dev)
        self.nn = torch.empty((n_max,), dtype=torch.int32, device=dev)
        self.cur = torch.empty((n_max, 3), dtype=torch.float32, device=dev)
        self.sums = torch.empty((PIN_GN_REPLICAS, PIN_GN_NSUMS), dtype=torch.float64, device=dev)
        self.sums_host = torch.empty((PIN_GN_REPLICAS, PIN_GN_NSUMS), dtype=torch.float64).pin_memory()
        self.on_knn = None  # optional hooks(start: bool) used by bench.py to bracket the kNN / GN launches
        self.on_gn = None
        self.fuse_solve = True  # pin_gn_accumulate_solve per iteration (False: pin_gn_accumulate_dev + pin_gn_solve)
        self.bricks = None  # ops.BrickCache built for (time_filtering, local) of the calls below
        self.state = self.state_host = None
        # device loop: Morton-order the source points once per registration (pin_spatial_sort); cell ~ voxel / 4
        self.sort_points, self.sort_cell, self.sort_min_points = True, max(float(st.resolution) / 4.0, 1e-3), 4096
        self._sorted = self._sort_ws = None
        # device loop, brick cache: searches coherent across the iterations (pin_gn_knn_coherent) -- a query that has not
        # left the margin of its last full search only re-ranks its k winners; bit-identical records.  OFF by default:
        # measured neutral on the C3 frame -- the pose keeps moving by more than the median margin
        # (1 cm) for most of the 50 iterations, and a wave only skips the full search when all its 8 queries stand still
        self.coherent = os.environ.get("PIN_KNN_COHERENT", "0") == "1"
        self._coh = None
        # device loop, brick cache: per-query candidate lists kept across the iterations (pin_gn_knn_listed) -- while a query
        # stays in its voxel (40 cm against centimetres of motion) an iteration gathers the ~45 occupied candidate cells it
        # listed instead of deriving all 81 cells' entry offsets again; bit-identical records.  PIN_KNN_LISTED=0: plain search
        self.listed = os.environ.get("PIN_KNN_LISTED", "1") != "0"
        self._cells = None

    def step(self, src: torch.Tensor, T: Optional[np.ndarray], time_filtering=True, local=True, labels=None,
             color=None):
        n = src.shape[0]
        out = (self.nbr[:n], self.nn[:n], self.cur[:n])
        if color is not None:  # the tile kernel with the colour term copies both decoder images (the colour one: ops.color_term)
            self.fs.stage_decoder()
        if self.on_knn:
            self.on_knn(True)
        ops.knn_query(self.st, src, self.fs.k, time_filtering=time_filtering, local=local, pose=T, out=out,
                      bricks=self.bricks)
        if self.on_knn:
            self.on_knn(False)
        cur = out[2] if T is not None else src
        ops.gn_accumulate(self.fs, self.gp, cur, out[0], out[1], sdf_labels=labels, sums=self.sums, color=color)
        self._sums_clean = False
        self.sums_host.copy_(self.sums, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        sums = self.sums_host.numpy()
        if not np.isfinite(sums).all():  # (the host-driven step has no state read-back: ask for the status word only when the sums are bad)
            ops.raise_on_status(ops.status())
            raise RuntimeError("registration_step: non-finite normal-equation sums (non-finite source points, map or decoder)")
        return ops.solve_gn(sums, self.lm_lambda)

    def _sort_into_buffer(self, src: torch.Tensor, stream):
        L = _lib.lib()
        n = src.shape[0]
        if self._sorted is None or self._sorted.shape[0] < n:
            self._sorted = torch.empty((n, 3), dtype=torch.float32, device=src.device)
            nb = int(L.pin_maint_workspace_bytes(n))
            self._sort_ws = torch.empty((nb,), dtype=torch.uint8, device=src.device)
        check(L.pin_spatial_sort(src.data_ptr(), n, float(self.sort_cell), self._sorted.data_ptr(), None, self._sort_ws.data_ptr(),
                                 self._sort_ws.numel(), stream), "pin_spatial_sort")

    def presort(self, src: torch.Tensor, stream: Optional[torch.cuda.Stream] = None):
        """The Morton ordering of track()'s source points ahead of time, on `stream` (default: the current one) -- a loader
        that prepares frame f+1's scan on its own stream while the main stream trains on frame f queues this behind the scan
        chain: 60 us of launches less between Mapper.mapping and the first registration iteration.  The next track() call on
        the SAME tensor (address, length) waits for the order through an event instead of sorting; any other call sorts as
        usual.  The caller vouches that no registration of this tracker is still running (track() returns after its read-back)."""
        if not (self.sort_points and src.is_cuda and src.shape[0] >= self.sort_min_points):
            return
        s = stream if stream is not None else torch.cuda.current_stream(src.device)
        with torch.cuda.stream(s):
            self._sort_into_buffer(src, ops._stream())
            if getattr(self, "_presort_ev", None) is None:
                self._presort_ev = torch.cuda.Event()
            self._presort_ev.record(s)
        self._presorted = ((src.data_ptr(), src.shape[0], float(self.sort_cell)), self._presort_ev)

    def track(self, src: torch.Tensor, T_init: np.ndarray, iters: int, term_deg: float = 0.01,
              term_m: float = 0.001, early_exit: bool = True, min_valid_ratio: float = 0.2,
              time_filtering=True, local=True, labels=None, color=None, probe: Optional[list] = None):
        """Device-resident GN loop (Tracker.tracking, tracker.py:114-184): `iters` x (kNN with the
        pose read from device state, fused SDF+Jacobian+sums, one-wave 6x6 solve + loop control)
        enqueued back to back; kernels turn into no-ops once the loop has ended on the device.
        ONE read-back per call.  Returns (T, valid_count, residual_cm, iterations, valid_flag,
        extra) with extra = dict(N_raw, mse, converged)."""
        L = _lib.lib()
        n = src.shape[0]
        stream = ops._stream()
        # The loop sums over the source points, so their order is free; the down-sampler hands them over ordered by an
        # x-fastest voxel id, and both per-iteration kernels are ~20 % faster on a Morton-ordered scan (30 -> 25 us
        # kNN, 37 -> 34.5 us GN per 98.7k points, scripts/knn_order_probe.py): one sort per registration pays after
        # the second iteration.  Per-point inputs (labels, colours) would have to follow: those calls keep the order.
        pre, self._presorted = getattr(self, "_presorted", None), None
        if self.sort_points and labels is None and color is None and n >= self.sort_min_points and iters > 2:
            if pre is not None and pre[0] == (src.data_ptr(), n, float(self.sort_cell)):
                torch.cuda.current_stream().wait_event(pre[1])  # presort(): the order is there already, queued on another stream
            else:
                self._sort_into_buffer(src, stream)
            src = self._sorted[:n]
        if self.state is None:
            self.state = torch.empty(_lib.PIN_GN_STATE_DOUBLES, dtype=torch.float64, device=src.device)
            self.state_host = torch.empty(_lib.PIN_GN_STATE_DOUBLES, dtype=torch.float64).pin_memory()
        T0 = np.ascontiguousarray(np.asarray(T_init, dtype=np.float64))
        lp = _lib.GnLoopParams()
        lp.lm_lambda, lp.term_thre_deg, lp.term_thre_m = float(self.lm_lambda), float(term_deg), float(term_m)
        lp.min_valid_ratio, lp.max_increment_ratio, lp.min_valid_points = float(min_valid_ratio), 1.1, 30
        lp.iter_n, lp.early_exit = int(iters), int(bool(early_exit))
        # (the loop parameters go INTO the state: pin_gn_accumulate_solve is then two launches per iteration -- the tile kernel's
        # last block runs the solve)
        check(L.pin_gn_loop_init(self.state.data_ptr(), T0.ctypes.data, n, C.byref(lp), stream), "pin_gn_loop_init")
        # the solve kernel of every iteration leaves the sums zeroed for the next one -- also the last one of the previous call:
        # a fill launch is needed only the first time and after the host-driven step(), which does not run that kernel
        if not getattr(self, "_sums_clean", False):
            self.sums.zero_()
        self._sums_clean = False  # (until this call's last solve has run: set again behind the read-back below)
        sp = self.st.params(time_filtering=time_filtering, local=local)
        # the decoder does not change during a registration: stage it once for all launches (pin_stage_decoder; the
        # colour decoder was staged by ops.color_term)
        self.fs.stage_decoder()
        f = self.fs.params()
        bc = None
        if self.bricks is not None:
            if self.bricks.mode[:2] != (bool(time_filtering), bool(local)):
                raise RuntimeError("brick cache was built for another query mode")
            bc = self.bricks.params()
        sp_r, f_r, gp_r, lp_r = C.byref(sp), C.byref(f), C.byref(self.gp), C.byref(lp)
        ct_r = C.byref(color) if color is not None else None
        bc_r = C.byref(bc) if bc is not None else None
        src_p, cur_p, nbr_p, nn_p = src.data_ptr(), self.cur.data_ptr(), self.nbr.data_ptr(), self.nn.data_ptr()
        sums_p, st_p = self.sums.data_ptr(), self.state.data_ptr()
        lab_p = None if labels is None else labels.data_ptr()
        k = self.fs.k
        coh = self.coherent and bc is not None
        if coh:
            if self._coh is None or self._coh[0].shape[0] < n:
                rows = self.nbr.shape[0]
                self._coh = (torch.empty((rows, 4), dtype=torch.float32, device=src.device),
                             torch.empty((rows, 8), dtype=torch.int32, device=src.device))
            cs_p, cw_p = (t.data_ptr() for t in self._coh)
        listed = self.listed and bc is not None and not coh
        if listed:
            stride = int(L.pin_knn_list_stride(int(sp.n_cand)))
            if self._cells is None or self._cells[0].shape[0] < n or self._cells[1].shape[1] != stride:
                rows = self.nbr.shape[0]
                self._cells = (torch.empty((rows, 4), dtype=torch.int32, device=src.device),
                               torch.empty((rows, stride), dtype=torch.int32, device=src.device))
            cell_p, list_p = (t.data_ptr() for t in self._cells)
        iters_run = iters
        if os.environ.get("PIN_GN_SPLIT", "0") == "1" and listed and n >= 8192 and lab_p is None and ct_r is None and probe is None:
            # EXPERIMENT (measured and not kept, DESIGN section 8 "Round 6"): the Morton-ordered scan in two
            # halves on two streams -- the search of half B beside the tile kernel of half A.  Per iteration two cross-stream
            # dependencies (solve -> search B, tile kernel B -> solve).
            main = torch.cuda.current_stream()
            if getattr(self, "_split_stream", None) is None:
                self._split_stream = torch.cuda.Stream(device=src.device)
                self._split_ev = [torch.cuda.Event() for _ in range(3)]
            side, (ev_solved, ev_b, ev_start) = self._split_stream, self._split_ev
            side_h = side.cuda_stream
            nA = ((n // 2) + 63) & ~63
            nB = n - nA
            off = lambda p, rows, width: p + 4 * rows * width  # noqa: E731 (float32 / int32 rows)
            srcB, curB, nbrB, nnB = off(src_p, nA, 3), off(cur_p, nA, 3), off(nbr_p, nA, 4 * k), off(nn_p, nA, 1)
            cellB, listB = off(cell_p, nA, 4), off(list_p, nA, stride)
            ev_start.record(main)
            side.wait_event(ev_start)
            for it in range(iters):
                first = int(it == 0)
                rc = L.pin_gn_knn_listed(sp_r, bc_r, src_p, nA, k, st_p, cur_p, nbr_p, nn_p, cell_p, list_p, first, stream)
                rc |= L.pin_gn_knn_listed(sp_r, bc_r, srcB, nB, k, st_p, curB, nbrB, nnB, cellB, listB, first, side_h)
                rc |= L.pin_gn_accumulate_dev(f_r, gp_r, ct_r, cur_p, nbr_p, nn_p, None, nA, sums_p, st_p, stream)
                rc |= L.pin_gn_accumulate_dev(f_r, gp_r, ct_r, curB, nbrB, nnB, None, nB, sums_p, st_p, side_h)
                ev_b.record(side)
                main.wait_event(ev_b)
                rc |= L.pin_gn_solve(sums_p, st_p, lp_r, stream)
                ev_solved.record(main)
                side.wait_event(ev_solved)
                if rc:
                    check(rc, "split registration iteration")
            iters = 0  # (the loop below is skipped)
        for it in range(iters):
            if self.on_knn:
                self.on_knn(True)
            if listed:  # iteration 0 builds every query's candidate list; later ones run off it while the query keeps its voxel
                rc = L.pin_gn_knn_listed(sp_r, bc_r, src_p, n, k, st_p, cur_p, nbr_p, nn_p, cell_p, list_p, int(it == 0), stream)
            elif coh:  # iteration 0: the full search that leaves the coherent state; later ones use it
                rc = L.pin_gn_knn_coherent(sp_r, bc_r, src_p, n, k, st_p, cur_p, nbr_p, nn_p, cs_p, cw_p, it, stream)
            else:
                rc = L.pin_gn_knn(sp_r, bc_r, src_p, n, k, st_p, cur_p, nbr_p, nn_p, stream)
            if self.on_knn:
                self.on_knn(False)
            if self.on_gn:
                self.on_gn(True)
            if self.fuse_solve:  # the tile kernel's last block runs the solve (a state from pin_gn_loop_init): one launch
                rc |= L.pin_gn_accumulate_solve(f_r, gp_r, ct_r, lp_r, cur_p, nbr_p, nn_p, lab_p, n, sums_p, st_p, stream)
            else:  # tile kernel and solve kernel, one by one (tests: both forms agree)
                rc |= L.pin_gn_accumulate_dev(f_r, gp_r, ct_r, cur_p, nbr_p, nn_p, lab_p, n, sums_p, st_p, stream)
                rc |= L.pin_gn_solve(sums_p, st_p, lp_r, stream)
            if self.on_gn:
                self.on_gn(False)
            if rc:
                check(rc, "pin_gn_knn / pin_gn_accumulate_solve")
            if probe is not None:  # diagnostics (bench.py --coherent-probe): synchronises every iteration
                torch.cuda.synchronize()
                T_now = self.state[:16].cpu().numpy().reshape(4, 4).copy()
                rec = dict(iteration=it, translation_step_m=None if not probe else float(np.linalg.norm(T_now[:3, 3] - probe[-1]["_T"][:3, 3])),
                           rotation_step_rad=None if not probe else float(np.linalg.norm(T_now[:3, :3] - probe[-1]["_T"][:3, :3])), _T=T_now)
                if coh:
                    stale = (self._coh[0][:n, :3] != self.cur[:n]).any(1)
                    rec["coherent_query_share"] = float(stale.float().mean().item())
                    rec["margin_m_median"] = float(self._coh[0][:n, 3].sqrt().median().item())
                    rec["margin_zero_share"] = float((self._coh[0][:n, 3] == 0).float().mean().item())
                probe.append(rec)
        self.state_host.copy_(self.state, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        self._sums_clean = iters_run > 0
        s = self.state_host.numpy()
        if s[_lib.PIN_GN_STATE_STATUS] != 0.0:  # sticky device flags the solve kernel copied into the read-back (pin_status)
            ops.raise_on_status(int(s[_lib.PIN_GN_STATE_STATUS]))
        if not np.isfinite(s[:18]).all():
            raise RuntimeError("Tracker.tracking: non-finite pose / residual in the registration state (non-finite source points, "
                               "map or decoder outputs)")
        T = s[:16].reshape(4, 4).copy()
        extra = dict(N_raw=s[24:60].reshape(6, 6).copy(), mse=float(s[23]), converged=bool(s[20]))
        return T, int(s[18]), float(s[17]), int(s[22]), bool(s[19]), extra


class MapTrainer:
    def __init__(self, st: ops.SearchState, fs: ops.FieldState, pool_coord, pool_label, pool_weight, pool_ts,
                 ts_update, *, bs: int, decimation: int, sigma: float, weight_e: float, eik_eps: float,
                 lr: float = 0.01, adam_eps: float = 1e-15, loss_weight_on: bool = False, train_decoder: bool = True,
                 eikonal: bool = True, rank: int = 0, world: int = 1, comm=None, dp_mode: str = "spatial"):
        self.st, self.fs = st, fs
        self.pool = (pool_coord, pool_label, pool_weight, pool_ts)
        self.ts_update = ts_update
        self.bs, self.dec = int(bs), int(decimation)
        self.sigma, self.weight_e, self.eik_eps = sigma, weight_e, eik_eps
        self.lr, self.adam_eps, self.loss_weight_on = lr, adam_eps, loss_weight_on
        self.train_decoder = train_decoder
        self.rank, self.world = rank, world
        # data-parallel: the transport of the two exchanges (collective.RcclComm: RCCL through the C ABI).  A
        # communicator of one rank is allowed (the same code path, used to exercise RCCL on a single-GPU box).
        self.comm = comm
        if world > 1 and comm is None:
            raise ValueError("MapTrainer(world > 1) needs a collective (pin_slam_amd.collective.RcclComm)")
        # how the batch is cut over the ranks: "spatial" (pin_slam_amd.dp: k-d boxes of the voxel grid, lazy Adam on the
        # owned rows, one all-reduce of [decoder | halo rows] per iteration) or "dense" (contiguous index shards, one
        # all-reduce of the whole [decoder | feature] gradient per iteration, replicated dense Adam)
        if dp_mode not in ("spatial", "dense"):
            raise ValueError("dp_mode: spatial | dense")
        self.dp_mode = dp_mode if comm is not None else None
        self.dp = None
        self.overlap_weight_grad = None  # None = automatic, True / False force (see step_batch)
        self.defer_dec_reduce = False   # see step_batch; Mapper.mapping switches it on for its loop
        self.lazy_finished = False
        self._pending_partial, self._iters_planned = None, 0
        self._wg_stream, self._wg_ev, self._wg_pending = None, None, False
        # spatial shards: the all-reduce of iteration i on a side stream, beside the weight gradient of iteration i and the
        # lazy-Adam launch of iteration i + 1 (step_batch); PIN_DP_OVERLAP=0 keeps everything on one stream
        self.overlap_exchange = os.environ.get("PIN_DP_OVERLAP", "1") != "0"
        # a call whose neighbour records are per pool sample applies the training-mode side effects (certainty += w, ts_update max)
        # ONCE at its end from the draw counts (pin_count_draws + pin_certainty_from_records) instead of with k atomics per query
        # and iteration in the tile kernel: set per call by the Mapper (deferred_side_effects), PIN_DEFER_CERTAINTY=0 turns it off
        self.defer_side_effects = False
        self._dp_stream, self._dp_ev, self._dp_pending = None, None, None
        self.on_grads = None  # optional hook(flat gradient buffer) between the all-reduce and the optimiser step
        self.on_allreduce = None  # optional hook(start: bool) around the gradient exchange (bench.py brackets it with events)
        self.on_color_grads = None  # optional hook(flat colour gradient payload) behind the dense shards' colour exchange (tests)
        self._cert0 = self._cert_scratch = None
        dev = fs.feats.device
        self._store = None
        self.resize(fs)
        from .sharding import n_eik_global, shard_range
        if self.dp_mode == "spatial":
            if eikonal == "analytic":
                raise NotImplementedError("spatially sharded mapper with the analytic Eikonal term (use dp_mode='dense')")
            from .dp import SpatialShards
            self.dp = SpatialShards(rank, world, comm, dev)
            self.bs_local = self.bs  # (capacity of the per-rank gather buffers is set per call, _shard_buffers)
            self.buf = None
        else:
            assert self.bs % world == 0, "global batch must divide over the ranks"
            self.bs_local = self.bs // world
            start, _ = shard_range(self.bs, rank, world)
            # one gather + one kNN launch per GROUP of iterations on one GPU (their inputs do not depend on the training)
            q_iter = self.bs_local + 6 * (0 if eikonal in (False, "analytic") else (self.bs_local + self.dec - 1) // self.dec)
            group = max(1, min(16, (1 << 22) // max(q_iter, 1))) if world == 1 else 1
            self.buf = ops.TrainBuffers(self.bs_local, self.dec, fs.k, fs.hidden, fs.levels, eikonal=eikonal,
                                        shard_start=start, weighted_first=fs.weighted_first, group=group)
        self.coord = torch.empty((self.bs_local, 3), dtype=torch.float32, device=dev)
        self.label = torch.empty((self.bs_local,), dtype=torch.float32, device=dev)
        self.weight = torch.empty((self.bs_local,), dtype=torch.float32, device=dev)
        self.ts = torch.empty((self.bs_local,), dtype=torch.int32, device=dev)
        # Eikonal samples are the global batch's coord[::dec]; each rank owns those in its shard
        self.n_eik_global = (self.bs if eikonal == "analytic" else n_eik_global(self.bs, self.dec)) if eikonal else 0
        self.eikonal = eikonal
        self.total_iter = 0
        self.bricks = None
        self.fc = None  # colour field (set_color)
        self.fsem = None  # semantic field (set_semantic)

    def resize(self, fs: ops.FieldState):
        """Point the optimiser buffers at `fs` (the local map changes size every frame): views of
        capacity-managed stores, reallocated only when the map outgrows them.  Contents are
        undefined until reset_optimizer(), which every Mapper.mapping call starts with."""
        nf, nd, rows = fs.feats.numel(), fs.dec.numel(), fs.feats.shape[0]
        if self._store is None or self._store[0].numel() < nd + nf or self._nd != nd:
            cap = nd + int(nf * 1.25) + 1024
            dev = fs.feats.device
            self._store = tuple(torch.zeros((cap,), dtype=torch.float32, device=dev) for _ in range(3)) + (
                torch.zeros((cap // 8 + 8,), dtype=torch.uint8, device=dev),)
            self._nd = nd
        g, m, v, d = self._store
        # one flat gradient buffer [decoder | features]: a single all-reduce payload
        self.grad, self.m, self.v = g[:nd + nf], m[:nd + nf], v[:nd + nf]
        self.gdec, self.gfeat = self.grad[:nd], self.grad[nd:]
        # rows touched since the optimiser state was reset: Adam skips the others, bit-identically (ops.adam_step_rows)
        self.dirty = d[:rows]
        self.fs = fs
        if not hasattr(self, "lazy"):  # single GPU: rows are advanced only when an iteration reads them (ops.LazyAdam)
            self.lazy = ops.LazyAdam(self.lr, eps=self.adam_eps)
            self.lazy_c = ops.LazyAdam(self.lr, eps=self.adam_eps)
            self.lazy_on = False

    def set_color(self, fc: Optional[ops.FieldState], surface_range: float = 0.0, weight_i: float = 0.0,
                  train_decoder: bool = True):
        """Enable the colour branch of Mapper.mapping (mapper.py:668-671, 802-812): a second
        decoder with 3 sigmoid heads over the colour feature table, L1 loss on the surface
        samples, reusing the neighbour records / IDW weights of the geometry pass."""
        if fc is None:
            self.fc = None
            return
        # (dp_mode = 'dense': the colour table's gradient is a second whole-table payload [colour decoder | colour features] and
        # the batch's surface-sample count a third, tiny one -- step_batch)
        nf, nd = fc.feats.numel(), fc.dec.numel()
        if self.fc is None or self.cgrad.numel() != nf + nd:
            self.cgrad = torch.zeros((nd + nf,), dtype=torch.float32, device=fc.feats.device)
            self.cm, self.cv = torch.zeros_like(self.cgrad), torch.zeros_like(self.cgrad)
        self.fc, self.c_range, self.c_weight, self.c_train_dec = fc, float(surface_range), float(weight_i), train_decoder
        self.cgdec = self.cgrad[:nd]  # (spatial shards: re-pointed at the exchange buffer by plan_shards)

    def set_semantic(self, fsem: Optional[ops.FieldState], heads: int = 0, weight_s: float = 0.0, decimation: int = 1,
                     freespace_label_on: bool = False, train_decoder: bool = True):
        """Enable the semantic branch of Mapper.mapping (mapper.py:664-667, 782-800): a decoder with `heads` outputs over the
        geometry features (fsem.feats is fs.feats), log-softmax, NLL on the labelled samples.  Its feature gradients add into
        the geometry gradient buffer (one optimiser over both terms, as the reference's single backward), the decoder has its
        own dense Adam state."""
        if fsem is None:
            self.fsem = None
            return
        if self.comm is not None:
            raise NotImplementedError("semantic training on the data-parallel mapper (no shipped configuration; run it on one GPU)")
        nd = fsem.dec.numel()
        if self.fsem is None or self.sgdec.numel() != nd:
            dev = fsem.feats.device
            self.sgdec = torch.zeros((nd,), dtype=torch.float32, device=dev)
            self.sm, self.sv = torch.zeros_like(self.sgdec), torch.zeros_like(self.sgdec)
            self._sem_sel = None
        self.fsem, self.s_heads, self.s_weight, self.s_dec = fsem, int(heads), float(weight_s), max(1, int(decimation))
        self.s_freespace, self.s_train_dec = bool(freespace_label_on), bool(train_decoder)

    def _semantic_step(self, sem_label, step: int):
        """The semantic term of the iteration just run by train_step (same queries / records), then the decoder's Adam step."""
        n = self.buf.n_main
        if sem_label is None:
            raise RuntimeError("semantic_on but the batch carries no semantic labels (Mapper.process_frame needs frame labels)")
        if self._sem_sel is None or self._sem_sel[0].shape[0] < n:
            dev = self.fsem.feats.device
            self._sem_sel = (torch.empty((max(n, self.bs_local),), dtype=torch.uint8, device=dev), torch.empty((1,), dtype=torch.int32, device=dev))
        sel, cnt = ops.sem_select(sem_label[:n], self.s_freespace, self.s_dec, out=(self._sem_sel[0][:n], self._sem_sel[1]))
        self.sem_loss = ops.train_sem_step(self.fsem, self.buf, sem_label[:n], sel, cnt, self.gfeat, self.sgdec if self.s_train_dec else None,
                                           heads=self.s_heads, weight_s=self.s_weight)
        self.sem_count = cnt
        if self.s_train_dec:
            ops.adam_step(self.fsem.dec, self.sgdec, self.sm, self.sv, step, self.lr, eps=self.adam_eps)

    # ------------------------------------------------------------------ spatially sharded data-parallel mapping (dp.py)
    def plan_shards(self, pool_coord, hist, new, new_idx, num_nei_cells: int, pool_rows=None, pool_label=None, reuse_records=False):
        """Start of a spatially sharded Mapper.mapping call, after reset_optimizer(): boxes, halo, this rank's samples of
        every drawn batch (SpatialShards.plan) and buffers of the size that came out."""
        dp, fs = self.dp, self.fs
        eik = bool(self.eikonal)
        res = float(self.st.resolution)
        reach = int(num_nei_cells) + (int(np.ceil(float(self.eik_eps) / res - 1e-9)) if eik else 0)
        nd = self.m.numel() - fs.feats.numel()
        cnd = self.fc.dec.numel() if self.fc is not None else 0
        dp.plan(pool_coord, hist, new, new_idx, decimation=self.dec, eikonal=eik, resolution=res, reach=reach, pos=fs.pos,
                lazy_pending=self.lazy.state if self.lazy_on else None, nd=nd + cnd, pool_rows=pool_rows,
                color_pending=self.lazy_c.state if (self.fc is not None and self.lazy_on) else None,
                pool_label=pool_label, surface_range=getattr(self, "c_range", 0.0), reuse_records=bool(reuse_records))
        self.gdec = dp.xbuf[:nd]  # the decoder gradients live at the head of the exchange buffer
        if self.fc is not None:
            self.cgdec = dp.xbuf[nd:nd + cnd]
        b = self.buf
        if b is None or b.cap_main < dp.cap or b.cap_eik < dp.eik_cap or (dp.eik_cap == 0) != (b.cap_eik == 0):
            q_iter = dp.cap + 6 * dp.eik_cap
            group = max(1, min(16, (1 << 22) // max(q_iter, 1)))
            self.buf = ops.TrainBuffers(dp.cap, 1, fs.k, fs.hidden, fs.levels, eikonal=eik, weighted_first=fs.weighted_first,
                                        group=group, n_eik=dp.eik_cap)
            dev, G, cap = fs.feats.device, group, dp.cap
            self._shard_out = dict(coord=torch.empty((G, cap, 3), dtype=torch.float32, device=dev),
                                   label=torch.empty((G, cap), dtype=torch.float32, device=dev),
                                   weight=torch.empty((G, cap), dtype=torch.float32, device=dev),
                                   ts=torch.empty((G, cap), dtype=torch.int32, device=dev), color=None)
        if self.fc is not None and (self._shard_out.get("color") is None or self._shard_out["color"].shape[:2] != self._shard_out["label"].shape):
            self._shard_out["color"] = torch.empty(tuple(self._shard_out["label"].shape) + (3,), dtype=torch.float32, device=fs.feats.device)
        # (the partition lists use dp.cap / dp.eik_cap as strides; the buffers above may be larger: gather with the lists' strides)
        if self.buf.cap_main != dp.cap or self.buf.cap_eik != dp.eik_cap:
            raise RuntimeError("shard buffer strides out of step with the partition")  # (grown together above)
        return dp.stats

    def run_shards(self, pool: dict, global_coord: bool, iters: int, on_iteration=None):
        """The iterations of a spatially sharded call: per group one gather launch, per iteration kNN + step_batch."""
        dp, buf = self.dp, self.buf
        out = self._shard_out
        reuse = dp.n_own is not None and dp.n_own > 0
        self.defer_side_effects = False
        if reuse:  # one search over this rank's pool samples for the whole call (the neural points do not move while the map trains)
            rec_nbr, rec_nn = dp.records(self.fs.k)
            ops.knn_query(self.st, dp.own_coord[:dp.n_own], self.fs.k, out=(rec_nbr, rec_nn, None), bricks=self.bricks)
            self.begin_deferred_side_effects(dp._hist, dp._new, dp._new_idx, dp.pool_rows)
        for it0 in range(0, iters, buf.group):
            gn = min(buf.group, iters - it0)
            C_color = 3 if self.fc is not None else 0
            if C_color and (pool.get("color") is None or pool["color"].shape[1] != 3):
                raise RuntimeError("colour training needs a 3-channel colour pool")
            dp.gather(pool, global_coord, C_color, it0, gn, out if C_color else dict(out, color=None), buf.query_all, self.eik_eps)
            if reuse:
                dp.gather_records(it0, gn, self.fs.k, buf.nbr_all, buf.nn_all)
            for j in range(gn):
                it = it0 + j
                buf.set_counts(j, int(dp.n_main[it]), int(dp.n_eik[it]))
                nm = buf.n_main
                if nm and reuse:  # the samples' records were copied: only the probes are searched
                    if buf.n_eik:
                        ops.knn_query(self.st, buf.query[nm:], self.fs.k, out=(buf.nbr[nm:], buf.nn[nm:], None), bricks=self.bricks)
                elif nm:
                    ops.knn_query(self.st, buf.query, self.fs.k, out=(buf.nbr, buf.nn, None), bricks=self.bricks)
                self.step_batch(out["coord"][j, :nm], out["label"][j, :nm], out["weight"][j, :nm], out["ts"][j, :nm], it + 1,
                                color_label=out["color"][j, :nm] if C_color else None, queries_ready=True, knn_ready=True,
                                surface_count=dp.surf_counts[it:it + 1] if C_color else None)
                if on_iteration is not None:
                    on_iteration(it)
        if reuse:  # the side effects of this rank's samples, once (the other ranks' arrive through publish())
            self.apply_deferred_side_effects(rec_nbr, rec_nn, pool["ts"], dp.pool_rows, pool_to_rec=dp.pool_to_own)

    def iteration(self, index_local: torch.Tensor, step: int):
        ops.gather_batch(*self.pool, index_local, (self.coord, self.label, self.weight, self.ts))
        self.step_batch(self.coord, self.label, self.weight, self.ts, step)

    def knn_group(self, n_iters: int):
        """Neighbour search of the first `n_iters` iterations' queries of the group buffers in one launch."""
        n = n_iters * self.buf.Q
        ops.knn_query(self.st, self.buf.query_all[:n], self.fs.k, out=(self.buf.nbr_all[:n], self.buf.nn_all[:n], None),
                      bricks=self.bricks)

    def _two_streams(self) -> bool:
        """step_batch's `overlap`: the weight gradient and the decoder's step of an iteration on a side stream."""
        want = self.overlap_weight_grad if self.overlap_weight_grad is not None else (self.fc is not None)
        return bool(self.lazy_on and self.train_decoder and want and self.on_grads is None and self.dp is None
                    and (self.fs.weighted_first or self.fs.levels == 1) and os.environ.get("PIN_MLP", "") != "f32")

    def can_step_group(self, color_label=None) -> bool:
        """What pin_train_group_steps runs behind the ABI: the one-GPU training loop on the lazy exact Adam -- in line (the decoder
        riding along in the lazy launch, or frozen) or in step_batch's two-stream form (a trained decoder's weight gradient and step on
        a side stream: the default with a colour decoder), with or without the colour branch -- no semantic branch, no ranks, no hooks,
        no analytic Eikonal term."""
        if not (self.lazy_on and self.dp is None and self.comm is None and self.on_grads is None
                and self.on_allreduce is None and self.fsem is None and not self.buf.analytic and self.fs.dec_image is not None
                and os.environ.get("PIN_TRAIN_GROUP", "1") != "0"):
            return False
        return self.fc is None or (color_label is not None and color_label.shape[-1] == 3 and color_label.stride(-1) == 1
                                   and color_label.stride(-2) == 3 and self.fc.dec_image is not None)

    def step_group(self, label, weight, ts, first_step: int, n_iters: int, color_label=None):
        """Iterations first_step .. first_step + n_iters - 1 on the group buffers (queries, records and counts of iteration i in slot i
        of buf.query_all / nbr_all / nn_all, labels / weights / timestamps / colours in row i of the [group, bs] tensors) in ONE foreign
        call: per iteration the launches of step_batch(queries_ready=True, knn_ready=True), queued by a C loop."""
        buf, fs, nd = self.buf, self.fs, self.gdec.numel()
        side = not self.defer_side_effects
        two = self._two_streams()
        tp = _lib.TrainParams()
        tp.n_main, tp.n_eik, tp.loss_weight_on = buf.n_main, buf.n_eik, int(bool(self.loss_weight_on))
        tp.sigma, tp.weight_e, tp.eik_eps = float(self.sigma), float(self.weight_e), float(np.float32(self.eik_eps))
        tp.inv_n_main = 1.0 / float(self.bs or buf.n_main)
        tp.inv_n_eik = 1.0 / float(self.n_eik_global or max(buf.n_eik, 1))
        tp.eik_analytic, tp.dec_image_current, tp.defer_weight_grad = 0, 1, 0
        f = fs.params()
        g = _lib.TrainGroup()
        g.n_iters, g.first_step = int(n_iters), int(first_step)
        g.last_of_call = int(first_step + n_iters - 1 >= self._iters_planned)
        n_rec = buf.Q * fs.k
        g.rows_form = int(n_rec >= self.lazy.rows_form_ratio * fs.feats.shape[0])
        g.query, g.query_stride = buf.query_all.data_ptr(), 3 * buf.Q
        g.nbr, g.nbr_stride = buf.nbr_all.data_ptr(), 4 * fs.k * buf.Q
        g.nn, g.nn_stride = buf.nn_all.data_ptr(), buf.Q
        g.sdf_label, g.label_stride = label.data_ptr(), label.stride(0)
        if weight is not None:
            g.sample_weight, g.weight_stride = weight.data_ptr(), weight.stride(0)
        if ts is not None:
            g.sample_ts, g.ts_stride = ts.data_ptr(), ts.stride(0)
        if side:
            g.certainty_rw, g.ts_update_rw = fs.certainty.data_ptr(), self.ts_update.data_ptr()
        g.feat_grad, g.loss_out = self.gfeat.data_ptr(), buf.loss.data_ptr()
        g.workspace, g.workspace_bytes = buf.ws.data_ptr(), buf.ws.numel() * 4
        g.n_records = n_rec
        lz = self.lazy
        g.exp_avg, g.exp_avg_sq, g.pending = self.m[nd:].data_ptr(), self.v[nd:].data_ptr(), lz.state.data_ptr()
        g.row_flags, g.n_rows = lz.flags.data_ptr(), fs.feats.shape[0]
        g.coef, g.t_max, g.beta1, g.beta2, g.eps = lz.coef.data_ptr(), lz.t_max, lz.b1, lz.b2, lz.eps
        d = None
        if self.train_decoder:  # (frozen: no rider, no weight gradient -- every frame of a run after freeze_after_frame)
            d = ops.LazyAdam._dense(self._dense(fs, self.gdec, self.m[:nd], self.v[:nd], True))
            g.dense, g.dec_grad = d, self.gdec.data_ptr()
        last = first_step + n_iters - 1
        if last > lz.t_max or first_step <= lz.t:
            raise ValueError("steps must grow within one optimiser lifetime and stay within the count reset() was sized for")
        keep = [f, tp, d]  # (ctypes structures the group points at: alive until the call has returned)
        if self._wg_pending:  # (the side work of an iteration stepped from Python, or of the group before)
            torch.cuda.current_stream().wait_event(self._wg_ev[1])
            self._wg_pending = False
        if two:
            if self._wg_stream is None:
                self._wg_stream = torch.cuda.Stream(device=fs.feats.device)
                self._wg_ev = (torch.cuda.Event(), torch.cuda.Event())
            g.side_stream = self._wg_stream.cuda_stream
        elif self.fc is None and self._pending_partial is not None:
            g.partial, g.partial_slots, _n, g.partial_scale = self._pending_partial
        if self.fc is not None:
            fc, lc, cnd = self.fc, self.lazy_c, self.fc.dec.numel()
            if last > lc.t_max or first_step <= lc.t:
                raise ValueError("colour table: steps must grow within one optimiser lifetime")
            ops.color_workspace(buf, fc)
            cf = fc.params()
            cp = _lib.TrainColorParams()
            cp.n_main, cp.loss_weight_on = buf.n_main, int(bool(self.loss_weight_on))
            cp.surface_range, cp.weight_i, cp.dec_image_current = float(self.c_range), float(self.c_weight), 1
            g.fc, g.cp = C.cast(C.pointer(cf), C.c_void_p), C.cast(C.pointer(cp), C.c_void_p)
            g.color_label, g.color_stride = color_label.data_ptr(), color_label.stride(0)
            g.c_feat_grad, g.c_loss_out = self.cgrad[cnd:].data_ptr(), buf.color_loss.data_ptr()
            g.c_workspace, g.c_workspace_bytes = buf.color_ws.data_ptr(), buf.color_ws.numel() * 4
            g.c_exp_avg, g.c_exp_avg_sq = self.cm[cnd:].data_ptr(), self.cv[cnd:].data_ptr()
            g.c_pending, g.c_row_flags = lc.state.data_ptr(), lc.flags.data_ptr()
            if self.c_train_dec:
                g.c_dec_grad = self.cgdec.data_ptr()
                cd = ops.LazyAdam._dense(self._dense(fc, self.cgdec, self.cm[:cnd], self.cv[:cnd], True))
                g.c_dense = cd
                keep.append(cd)
            keep += [cf, cp]
        check(_lib.lib().pin_train_group_steps(C.byref(f), C.byref(tp), C.byref(g), ops._stream()), "pin_train_group_steps")
        lz.t = last
        if g.rows_form:
            lz.rows_launches += n_iters
        if self.fc is not None:
            self.lazy_c.t = last
            if g.rows_form:
                self.lazy_c.rows_launches += n_iters
        if two:
            self._wg_ev[1].record(self._wg_stream)  # (the last iteration's weight gradient and decoder step: whoever needs them waits)
            self._wg_pending = True
            self._pending_partial = None
        else:
            self._pending_partial = (g.partial, g.partial_slots, d.n, g.partial_scale) if (g.partial and d is not None) else None
        self.total_iter += n_iters
        del keep

    def step_batch(self, coord, label, weight, ts, step: int, color_label=None, queries_ready: bool = False,
                   knn_ready: bool = False, surface_count=None, sem_label=None):
        """One iteration on an explicit (already gathered) batch shard.  queries_ready: buf.query already holds this
        batch's queries (written by the gather launch); knn_ready: buf.nbr / buf.nn hold their neighbours (knn_group)."""
        nd = self.gdec.numel()
        lazy = self.lazy_on
        # lazy exact Adam: ONE launch per iteration, before the forward pass -- the rows this iteration reads settle the
        # step they still owe from the iteration that last read them (+ the gradient-free steps since), and the decoder's
        # step of the previous iteration rides along in the same launch
        dense = self._dense(self.fs, self.gdec, self.m[:nd], self.v[:nd], lazy, self._pending_partial) if self.train_decoder else None
        self._pending_partial = None  # (this iteration's lazy launch takes it)
        # Two streams (one GPU, decoder training, fused tile paths): the weight gradient of this iteration, its finalize and the
        # decoder's step (with the image write-through) go to a side stream -- they touch the decoder, its gradient and
        # the workspace only -- while the caller's stream goes on with the colour branch and the lazy-Adam launch of
        # the NEXT iteration; the next tile kernel waits for both.  Measured: with a colour branch to overlap with
        # (C5) mapping 1.84 -> 1.73 ms per frame; without one (C3) the two cross-stream dependencies per iteration cost
        # more than the ~15 us they can hide (1.22 -> 1.44 ms) -- hence the default (None = only with a colour branch).
        want = self.overlap_weight_grad if self.overlap_weight_grad is not None else (self.fc is not None)
        overlap = bool(lazy and self.train_decoder and want and self.on_grads is None and self.dp is None
                       and (self.fs.weighted_first or self.fs.levels == 1) and os.environ.get("PIN_MLP", "") != "f32")
        if overlap:
            main = torch.cuda.current_stream()
            if self._wg_stream is None:
                self._wg_stream = torch.cuda.Stream(device=self.fs.feats.device)
                self._wg_ev = (torch.cuda.Event(), torch.cuda.Event())

            def pre():
                self.lazy.prepare(self.buf.nbr, self.fs.feats, self.gfeat, self.m[nd:], self.v[nd:], step, dense=None)
                if self._wg_pending:
                    main.wait_event(self._wg_ev[1])
                    self._wg_pending = False
        else:
            pre = (lambda: self.lazy.prepare(self.buf.nbr, self.fs.feats, self.gfeat, self.m[nd:], self.v[nd:], step, dense=dense)) if lazy else None
        # Spatial shards.  Nothing of iteration i + 1 up to its forward pass depends on the exchange of iteration i except
        # through the halo rows and the decoder: the all-reduce goes to a side stream -- first the halo rows (ready behind the tile
        # kernel), then the decoder gradients (ready behind the weight-gradient launch, which therefore runs BESIDE the halo
        # message) -- and the caller's stream goes on with the next iteration's search and its lazy-Adam launch (owned rows
        # only: the decoder's step no longer rides in it); the halo rows' step and the decoder's step follow the all-reduce
        # right in front of the next tile kernel (_dp_finish_exchange).  Same kernels, same operands, same order per datum.
        dp_overlap = bool(self.dp is not None and lazy and self.overlap_exchange and self.on_grads is None)
        dp_defer = False
        if dp_overlap:
            main = torch.cuda.current_stream()
            if self._dp_stream is None:
                self._dp_stream = torch.cuda.Stream(device=self.fs.feats.device)
                self._dp_ev = (torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event())
            dp_defer = bool(self.train_decoder and (self.fs.weighted_first or self.fs.levels == 1) and coord.shape[0] > 0
                            and os.environ.get("PIN_MLP", "") != "f32")

            def pre():
                self.lazy.prepare(self.buf.nbr, self.fs.feats, self.gfeat, self.m[nd:], self.v[nd:], step, dense=None)
                self._dp_finish_exchange()
        # One launch less per iteration (Mapper.mapping sets defer_dec_reduce): the weight-gradient launch leaves the decoder's gradient
        # as slot copies, and instead of a reduction launch the NEXT iteration's lazy launch -- whose tail blocks take the decoder's
        # step anyway -- sums them where it needs them (same sum, same order: the same bits).  Not on the call's last iteration
        # (its loss sums are the ones a caller can see, and the flush wants a plain gradient), nor with anything else that reads or
        # writes the decoder gradient between the launches (hooks, a second stream, ranks, colour / semantic branches).
        defer_reduce = bool(self.defer_dec_reduce and lazy and self.train_decoder and not overlap and self.dp is None and self.comm is None
                            and self.on_grads is None and self.fc is None and self.fsem is None and step < self._iters_planned
                            and coord.shape[0] > 0)
        if self.dp is not None and coord.shape[0] == 0:
            # none of this batch's samples fell into this rank's box: its rows settle nothing, the decoder still takes its
            # step (the dense rider of the lazy launch) and the exchange below still runs -- the other ranks wait in it
            if pre is not None:
                pre()
        else:
            side = not self.defer_side_effects
            ops.train_step(self.st, self.fs, self.buf, coord, label, weight, ts,
                           self.fs.certainty if side else None, self.ts_update if side else None, self.gfeat,
                           self.gdec if self.train_decoder else None,
                           sigma=self.sigma, weight_e=self.weight_e, eik_eps=self.eik_eps,
                           loss_weight_on=self.loss_weight_on, global_n_main=self.bs, global_n_eik=self.n_eik_global,
                           bricks=self.bricks, before_forward=pre, queries_ready=queries_ready, image_current=lazy,
                           knn_ready=knn_ready, defer_weight_grad=overlap or dp_defer, defer_dec_reduce=defer_reduce)
            if defer_reduce:
                self._pending_partial = ops.train_deferred_partial()  # (None: a path that reduced as usual)
        if overlap:
            self._wg_ev[0].record(main)
            side = self._wg_stream
            side.wait_event(self._wg_ev[0])
            with torch.cuda.stream(side):
                ops.train_weight_grad(self.buf, self.gdec)
                self.lazy.step_dense(dense, step)
                self._wg_ev[1].record(side)
            self._wg_pending = True
        if self.fsem is not None and coord.shape[0] > 0:  # semantic branch: adds into the geometry feature gradients of this iteration
            self._semantic_step(sem_label, step)
        if self.fc is not None:
            cnd = self.fc.dec.numel()
            cdense = self._dense(self.fc, self.cgdec, self.cm[:cnd], self.cv[:cnd], lazy) if self.c_train_dec else None
            dense_dp = self.comm is not None and self.dp is None  # contiguous index shards: whole-table exchange, replicated dense Adam
            if lazy:
                self.lazy_c.prepare(self.buf.nbr, self.fc.feats, self.cgrad[cnd:], self.cm[cnd:], self.cv[cnd:], step, dense=cdense)
            if dense_dp and surface_count is None:
                # color_diff_loss divides by the number of surface samples of the WHOLE batch (utils/loss.py:31-42): every rank
                # counts its shard, one one-word SUM exchange (exact in fp32: a count below 2^24), back to the int32 the kernel reads
                if getattr(self, "_surf_f", None) is None:
                    dev = self.fc.feats.device
                    self._surf_f = torch.zeros((1,), dtype=torch.float32, device=dev)
                    self._surf_i = torch.zeros((1,), dtype=torch.int32, device=dev)
                torch.sum(label.abs() < self.c_range, dim=(0,), keepdim=True, dtype=torch.float32, out=self._surf_f)
                self.comm.allreduce(self._surf_f, self._surf_f)
                self._surf_i.copy_(self._surf_f)
                surface_count = self._surf_i
            if coord.shape[0] > 0:  # (a rank whose box holds none of this batch's samples only takes the decoder's step above)
                ops.train_color_step(self.fc, self.buf, label, color_label, weight, self.cgrad[cnd:],
                                     self.cgdec if self.c_train_dec else None, surface_range=self.c_range,
                                     weight_i=self.c_weight, loss_weight_on=self.loss_weight_on, image_current=lazy,
                                     surface_count=surface_count, global_n_main=self.bs)
            if dense_dp:
                # SUM of the ranks' [colour decoder | colour features] gradients, then the same dense Adam step on every replica
                # (rows touched only by other ranks' shards arrive through the exchange) -- as the geometry table below
                payload = self.cgrad if self.c_train_dec else self.cgrad[cnd:]
                self.comm.allreduce(payload, payload)
                if self.on_color_grads is not None:
                    self.on_color_grads(payload)
                ops.adam_step(self.fc.feats, self.cgrad[cnd:], self.cm[cnd:], self.cv[cnd:], step, self.lr, eps=self.adam_eps)
                if self.c_train_dec:
                    ops.adam_step(self.fc.dec, self.cgdec, self.cm[:cnd], self.cv[:cnd], step, self.lr, eps=self.adam_eps)
            elif not lazy:
                ops.mark_rows(self.buf.nbr, self.dirty)  # the colour pass reuses the records of the geometry pass
                ops.adam_step_rows(self.fc.feats, self.cgrad[cnd:], self.cm[cnd:], self.cv[cnd:], self.dirty, step, self.lr,
                                   eps=self.adam_eps)
                if self.c_train_dec:
                    ops.adam_step(self.fc.dec, self.cgdec, self.cm[:cnd], self.cv[:cnd], step, self.lr, eps=self.adam_eps)
        if dp_overlap:
            dpx, side, (ev_a, ev_a2, ev_b) = self.dp, self._dp_stream, self._dp_ev
            color = None if self.fc is None else (self.fc.feats, self.cgrad[self.fc.dec.numel():])
            dpx.pack_halo(self.gfeat, color)
            ev_a.record(main)
            side.wait_event(ev_a)
            with torch.cuda.stream(side):
                dpx.allreduce_range(dpx.nd, dpx.nd + 8 * dpx.n_halo * dpx.tables, self.on_allreduce)
            if dp_defer:
                ops.train_weight_grad(self.buf, self.gdec)  # beside the halo message
            ev_a2.record(main)
            side.wait_event(ev_a2)
            with torch.cuda.stream(side):
                dpx.allreduce_range(0, dpx.nd)
                ev_b.record(side)
            self._dp_pending = (step, dense if self.train_decoder else None)
            self.total_iter += 1
            return
        if self.dp is not None:  # spatial shards: [decoder | halo rows] all-reduced, the halo rows' Adam step right behind
            self.dp.exchange(self.fs.feats, self.gfeat, step, self.lazy.coef, self.lazy.t_max, self.lazy.b1, self.lazy.b2,
                             self.lazy.eps, on_allreduce=self.on_allreduce,
                             color=None if self.fc is None else (self.fc.feats, self.cgrad[self.fc.dec.numel():]))
            if self.on_grads is not None:
                self.on_grads(self.dp.xbuf[:self.dp.nd + 8 * self.dp.n_halo * self.dp.tables])
            self.total_iter += 1
            return
        if self.comm is not None:  # SUM of the per-rank gradients of [decoder | features] (pin_allreduce_grads)
            if self.on_allreduce is not None:
                self.on_allreduce(True)
            self.comm.allreduce_grads(self.grad if self.train_decoder else self.gfeat)
            if self.on_allreduce is not None:
                self.on_allreduce(False)
        if self.on_grads is not None:
            self.on_grads(self.grad)
        if lazy:
            pass  # (this iteration's steps are taken by the next prepare() / by finish_optimizer())
        elif self.comm is None:
            ops.mark_rows(self.buf.nbr, self.dirty)
            ops.adam_step_rows(self.fs.feats, self.gfeat, self.m[nd:], self.v[nd:], self.dirty, step, self.lr, eps=self.adam_eps)
        else:  # rows touched by the other ranks' shards arrive through the all-reduce: dense update
            ops.adam_step(self.fs.feats, self.gfeat, self.m[nd:], self.v[nd:], step, self.lr, eps=self.adam_eps)
        if self.train_decoder and not lazy:
            ops.adam_step(self.fs.dec, self.gdec, self.m[:nd], self.v[:nd], step, self.lr, eps=self.adam_eps)
        self.total_iter += 1

    def begin_deferred_side_effects(self, hist, new, new_idx, pool_rows: int) -> bool:
        """Start of a call with per-pool-sample records: count how often every pool row is drawn (all iterations, one launch);
        the training launches then skip their certainty / ts atomics and apply_deferred_side_effects() does them once."""
        if os.environ.get("PIN_DEFER_CERTAINTY", "1") == "0" or self.fs.certainty is None:
            self.defer_side_effects = False
            return False
        dev = self.fs.feats.device
        c = getattr(self, "_draw_count", None)
        if c is None or c.numel() < pool_rows:
            c = self._draw_count = torch.zeros((int(pool_rows * 1.25) + 1024,), dtype=torch.int32, device=dev)
        else:
            c[:pool_rows].zero_()
        check(_lib.lib().pin_count_draws(hist.data_ptr(), hist.numel(), None if new is None else new.data_ptr(),
                                         None if new is None else new_idx.data_ptr(), 0 if new is None else new.numel(), c.data_ptr(),
                                         ops._stream()), "pin_count_draws")
        self.defer_side_effects = True
        return True

    def apply_deferred_side_effects(self, rec_nbr, rec_nn, pool_ts, pool_rows: int, pool_to_rec=None):
        """certainty[idx_k] += draws x w_k, ts_update[idx_k] = max(., sample ts) for every drawn pool sample, from its record."""
        if not self.defer_side_effects:
            return
        check(_lib.lib().pin_certainty_from_records(rec_nbr.data_ptr(), rec_nn.data_ptr(), self.fs.k,
                                                    None if pool_to_rec is None else pool_to_rec.data_ptr(), self._draw_count.data_ptr(),
                                                    None if pool_ts is None else pool_ts.data_ptr(), int(pool_rows), self.fs.certainty.data_ptr(),
                                                    self.ts_update.data_ptr(), ops._stream()), "pin_certainty_from_records")
        self.defer_side_effects = False

    def _dp_finish_exchange(self):
        """Behind the all-reduce of the last iteration (side stream): the halo rows' Adam step and the decoder's."""
        if self._dp_pending is None:
            return
        step_prev, dense_prev = self._dp_pending
        torch.cuda.current_stream().wait_event(self._dp_ev[2])
        color = None if self.fc is None else (self.fc.feats, self.cgrad[self.fc.dec.numel():])
        self.dp.halo_step(self.fs.feats, step_prev, self.lazy.coef, self.lazy.t_max, self.lazy.b1, self.lazy.b2, self.lazy.eps, color)
        if dense_prev is not None:
            self.lazy.step_dense(dense_prev, step_prev)
        self._dp_pending = None

    @staticmethod
    def _dense(fs: ops.FieldState, grad, m, v, lazy: bool, partial=None):
        """The decoder as the dense rider of the lazy optimiser, with its staged image when there is one; `partial`: the weight
        gradient the last training step left as slot copies (ops.train_deferred_partial), which the rider's step then sums itself."""
        if lazy and fs.dec_image is not None:
            d = (fs.dec, grad, m, v, fs.dec_image, fs.hidden, fs.levels, fs.out_dim)
        else:
            d = (fs.dec, grad, m, v)
        if partial is not None:
            d = d + (None,) * (8 - len(d)) + (partial,)
        return d

    def _stage_images(self):
        """One staging launch per Mapper.mapping call and decoder: from then on the lazy optimiser writes every parameter it
        updates through to the image (pin_adam_dense.image) and the training launches skip their staging kernel."""
        for name, fs in (("_img", self.fs), ("_img_c", self.fc)):
            if fs is None:
                continue
            if fs.dec_image is None:
                fs.dec_image = getattr(self, name, None)  # (a FieldState is rebuilt every frame: keep the buffer)
            fs.stage_decoder()
            setattr(self, name, fs.dec_image)

    def reset_optimizer(self, iters: Optional[int] = None):
        """setup_optimizer is called anew by every Mapper.mapping (mapper.py:615).  With the iteration count
        known (and one GPU) the feature tables use the lazy exact Adam: call finish_optimizer() after the last
        iteration; without it, the row-flagged / dense step."""
        # a call that ended between step_batch and finish_optimizer (an exception in the caller's loop) leaves the side
        # streams' hand-over state behind: order this stream behind whatever they still run and forget the owed steps --
        # they belong to the optimiser state that is thrown away below
        if self._wg_pending:
            torch.cuda.current_stream().wait_event(self._wg_ev[1])
            self._wg_pending = False
        if self._dp_pending is not None:
            # spatial shards: the halo rows' and the decoder's step of the aborted call's last iteration are owed on EVERY rank,
            # and the other ranks take them (their call did not fail): take them here as well -- the exchange itself has run,
            # all that is left is local -- so that the replicas of halo rows and decoder stay identical; the owned rows' lazy
            # steps are dropped with the optimiser state on every rank alike (they would have been settled by the flush of the
            # call that did not finish: the next call's publish() re-broadcasts every owned row)
            self._dp_finish_exchange()
        self.lazy_on = bool(iters) and (self.comm is None or self.dp is not None)
        self._pending_partial, self._iters_planned = None, int(iters or 0)  # (an aborted call's owed gradient goes with its optimiser state)
        self.lazy_finished = False
        if self.dp is not None and not self.lazy_on:
            raise ValueError("the spatially sharded mapper needs the iteration count (lazy Adam on the owned rows)")
        nd = self.gdec.numel()
        if self.lazy_on:
            dev = self.fs.feats.device
            self.m[:nd].zero_()
            self.v[:nd].zero_()
            self.lazy.reset(self.fs.feats.shape[0], iters, dev)
            self._stage_images()
        else:
            self.m.zero_()
            self.v.zero_()
            self.dirty.zero_()
        # after a lazy mapping() call the gradient buffer is already clean: every settled step clears its gradient and
        # the final flush settles every touched row and the decoder
        if not getattr(self, "_grad_clean", False):
            self.grad.zero_()
        self._grad_clean = False
        if self.fsem is not None:  # a new Adam per call (mapper.py:615) for the semantic decoder as well
            self.sm.zero_()
            self.sv.zero_()
            self.sgdec.zero_()
        if self.fc is not None:
            cnd = self.fc.dec.numel()
            if self.lazy_on:
                self.cm[:cnd].zero_()
                self.cv[:cnd].zero_()
                self.lazy_c.reset(self.fc.feats.shape[0], iters, self.fc.feats.device)
            else:
                self.cm.zero_()
                self.cv.zero_()
            self.cgrad.zero_()

    def finish_optimizer(self):
        """Bring the rows the lazy optimiser left behind to the final step (no-op otherwise)."""
        if not self.lazy_on:
            return
        nd = self.gdec.numel()
        dense = self._dense(self.fs, self.gdec, self.m[:nd], self.v[:nd], True, self._pending_partial) if self.train_decoder else None
        self._pending_partial = None
        if self._wg_pending:  # the side stream took the decoder through every step already (step_batch)
            torch.cuda.current_stream().wait_event(self._wg_ev[1])
            self._wg_pending = False
            dense = None
        if self._dp_pending is not None:  # (spatial shards, overlapped exchange: the last iteration's halo / decoder steps)
            self._dp_finish_exchange()
            dense = None
        stepped = self.lazy.t > 0
        self.lazy.flush(self.fs.feats, self.gfeat, self.m[nd:], self.v[nd:], dense=dense)
        if self.fc is not None:
            cnd = self.fc.dec.numel()
            cdense = self._dense(self.fc, self.cgdec, self.cm[:cnd], self.cv[:cnd], True) if self.c_train_dec else None
            self.lazy_c.flush(self.fc.feats, self.cgrad[cnd:], self.cm[cnd:], self.cv[cnd:], dense=cdense)
        if self.dp is not None and stepped:  # every rank's owned rows (and all side effects) everywhere; the moment arrays are free now
            self.dp.publish(self.fs.feats, self.m[nd:], self.fs.certainty, self._cert0, self._cert_scratch, self.ts_update,
                            color=None if self.fc is None else (self.fc.feats, self.cm[self.fc.dec.numel():]))
        self._grad_clean = self.train_decoder  # (a frozen decoder's gradient slot is never written either, but keep it simple)
        self.lazy_on = False
        self.lazy_finished = stepped  # (lazy.state now marks the rows this call's queries read: Mapper.mapping copies only those back)

    def mapping(self, index_batches):
        """One Mapper.mapping call: a fresh Adam state (mapper.py:615) and len(index_batches)
        iterations.  index_batches[i] is this rank's int32 shard of the i-th global batch."""
        self.reset_optimizer(len(index_batches))
        self.begin_side_effects()
        for i, idx in enumerate(index_batches):
            self.iteration(idx, i + 1)
        self.finish_optimizer()
        self.merge_side_effects()

    def begin_side_effects(self):
        """Data-parallel: remember the certainties before the call (pin_dp_cert_snapshot)."""
        if self.comm is None:
            return
        n = self.fs.certainty.shape[0]
        if self._cert0 is None or self._cert0.shape[0] < n:
            dev = self.fs.certainty.device
            self._cert0 = torch.empty(int(n * 1.25) + 1024, dtype=torch.float32, device=dev)
            self._cert_scratch = torch.empty_like(self._cert0)
        check(_lib.lib().pin_dp_cert_snapshot(self.fs.certainty.data_ptr(), self._cert0.data_ptr(), n,
                                              ops._stream()), "pin_dp_cert_snapshot")

    def merge_side_effects(self):
        """world > 1: certainty / ts_update side effects of the other ranks' shards, once per mapping call
        (pin_dp_sync_side_effects)."""
        if self.comm is None or self.dp is not None:  # (spatial shards: the side effects are merged with the rows, SpatialShards.publish)
            return
        self.comm.sync_side_effects(self.fs.certainty, self._cert0, self._cert_scratch, self.ts_update)
