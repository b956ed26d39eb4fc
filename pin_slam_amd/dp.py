"""Spatially sharded data-parallel mapper (kernels in csrc/dp.hip).

MI355X's xGMI is point-to-point (7 links x ~153 GB/s per GPU): a dense all-reduce of the 71 MB feature-gradient
table every iteration costs as much as a rank's share of the training step.  So the batch is cut by WHERE a sample
lies: the voxel grid is split into `world` boxes (a k-d split of the drawn batch) and rank r trains on the samples of
every drawn batch that fall in box r.  The gradient of a feature row well inside a box is then complete on its
owner, whose (lazy, exact) Adam step is the only one that row needs, and nobody else reads the row while the call
runs.  Only the rows within reach of a cut -- the halo, a few per cent of the map -- are shared: per iteration ONE
all-reduce of the compact buffer [decoder grads | halo-row grads] followed by the same dense Adam step on every
rank; at the end of Mapper.mapping every rank publishes the rows it owns with one more all-reduce.

The sum of the per-rank gradients is the reference's whole-batch gradient for ANY partition of the samples (the
losses are normalised by the global counts, utils/mapper.py:732-780), so the result does not depend on the boxes;
they only decide load balance and the size of the halo.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import numpy as np
import torch

from . import _lib, ops
from ._lib import DpRegions, check

INT_MIN, INT_MAX = -(1 << 31), (1 << 31) - 1


def kd_boxes(cells: np.ndarray, world: int) -> np.ndarray:
    """pin_dp_kd_boxes: [world][6] int32 boxes that tile the voxel grid (host code of libpinhip, no device work)."""
    cells = np.ascontiguousarray(cells, dtype=np.int32).reshape(-1, 3)
    out = np.empty((world, 6), dtype=np.int32)
    check(_lib.lib().pin_dp_kd_boxes(cells.ctypes.data, cells.shape[0], int(world), out.ctypes.data), "pin_dp_kd_boxes")
    return out


def kd_boxes_numpy(cells: np.ndarray, world: int) -> np.ndarray:
    """The same split in numpy (tests compare the two).  [world][6] int32 boxes (lo xyz, hi xyz exclusive; unbounded faces INT_MIN / INT_MAX) that tile the voxel grid:
    recursive splits of the sample cells `cells` [n,3] along the axis of largest extent, at the voxel coordinate
    that divides the samples in proportion to the ranks on either side.  Deterministic in its input (every rank
    calls it on the same sub-sample of the same drawn batch)."""
    cells = np.asarray(cells, dtype=np.int64).reshape(-1, 3)
    out = np.empty((world, 6), dtype=np.int64)

    def split(idx, first, count, lo, hi):
        if count == 1:
            out[first, :3], out[first, 3:] = lo, hi
            return
        c = cells[idx]
        n_left = count // 2
        if len(c) == 0:  # nothing to balance: an empty left box is legal
            axis = 0
            t = lo[0] if lo[0] != INT_MIN else (hi[0] if hi[0] != INT_MAX else 0)
        else:
            ext = c.max(0) - c.min(0)
            axis = int(np.argmax(ext))  # (first of equal extents)
            v = np.sort(c[:, axis], kind="stable")
            want = (len(v) * n_left) // count
            t = int(v[min(want, len(v) - 1)])  # cells < t go left
            # ties at the cut: of the two admissible thresholds take the one closer to the wanted share
            below, below_next = int(np.searchsorted(v, t, side="left")), int(np.searchsorted(v, t, side="right"))
            if abs(below_next - want) < abs(below - want):
                t += 1
            if ext[axis] > 0:  # keep occupied cells on both sides when there is a choice
                t = min(max(t, int(v[0]) + 1), int(v[-1]))
        hi_l, lo_r = list(hi), list(lo)
        hi_l[axis], lo_r[axis] = t, t
        left = idx[cells[idx, axis] < t] if len(idx) else idx
        right = idx[cells[idx, axis] >= t] if len(idx) else idx
        split(left, first, n_left, list(lo), hi_l)
        split(right, first + n_left, count - n_left, lo_r, list(hi))

    split(np.arange(len(cells)), 0, world, [INT_MIN] * 3, [INT_MAX] * 3)
    return out.astype(np.int32)


def region_of_cells(boxes: np.ndarray, cells: np.ndarray) -> np.ndarray:
    """Host mirror of the device lookup (tests)."""
    cells = np.asarray(cells, np.int64)
    r = np.full(len(cells), -1, np.int64)
    for i, b in enumerate(np.asarray(boxes, np.int64)):
        inside = np.all(cells >= b[:3], 1) & np.all(cells < b[3:], 1)
        r[inside & (r < 0)] = i
    return r


class SpatialShards:
    """Per-trainer state of the spatially sharded mapper: the boxes, the halo, the compact exchange buffer and the
    per-call partition of the drawn batches."""

    KD_SAMPLES = 8192

    def __init__(self, rank: int, world: int, comm, device):
        if not (1 <= world <= 64):
            raise ValueError("world size 1..64")
        self.rank, self.world, self.comm, self.device = rank, world, comm, device
        self.boxes_dev = torch.zeros((world, 6), dtype=torch.int32, device=device)
        self._boxes_host = torch.zeros((world, 6), dtype=torch.int32).pin_memory()
        self._cells = self._cells_host = None
        self.halo_rows = self.owner = self.xbuf = self.hm = self.hv = None
        self.n_halo = 0
        self._cnt = torch.zeros((1,), dtype=torch.int32, device=device)
        self.sel = self.esel = self.counts = None
        self.cap = self.ecap = 0
        self.counts_host = None
        self._ws = self._pool_region = self._halves = self._halves_host = None
        self.lists = self.offsets = self._send = self._recv = self._side = self.surf_counts = None
        self.own_coord = self.pool_to_own = self.rec = None  # record reuse (plan(reuse_records=True))
        self.own_cap, self.n_own = 0, None
        # end-of-call merge: "gather" = every rank publishes the rows it owns (all-gather, certainty / ts ride along, the halo's
        # side effects in compact form); "reduce" = all-reduce of the whole table + the side-effect all-reduces over every row
        self.merge = "gather"
        self.fixed_boxes: Optional[np.ndarray] = None  # tests may pin the partition
        self.stats = {}

    # ------------------------------------------------------------------ regions
    def regions(self, resolution: float, reach: int) -> DpRegions:
        rg = DpRegions()
        rg.boxes, rg.world, rg.rank, rg.reach = self.boxes_dev.data_ptr(), self.world, self.rank, int(reach)
        rg.resolution = float(np.float32(resolution))
        return rg

    def _draw_args(self, hist, new, new_idx):
        n_hist = hist.shape[1]
        n_new = 0 if new is None else new.shape[1]
        return (hist.data_ptr(), n_hist, None if new is None else new.data_ptr(), None if new is None else new_idx.data_ptr(),
                n_hist + n_new, n_hist, n_new)

    def plan(self, pool_coord: torch.Tensor, hist: torch.Tensor, new: Optional[torch.Tensor], new_idx, *, decimation: int,
             eikonal: bool, resolution: float, reach: int, pos: torch.Tensor, lazy_pending: Optional[torch.Tensor],
             nd: int, pool_rows: Optional[int] = None, color_pending: Optional[torch.Tensor] = None,
             pool_label: Optional[torch.Tensor] = None, surface_range: float = 0.0, reuse_records: bool = False):
        """Everything a Mapper.mapping call needs before its first iteration: boxes from a sub-sample of the first drawn
        batch (host, one small read-back), the halo of the feature rows at `pos`, the partition of ALL drawn batches
        (hist [iters][n_hist] / new [iters][n_new] int64, as Mapper._draw_all makes them) and its counts (second
        read-back).  nd = decoder parameters in front of the halo gradients in the exchange buffer."""
        L = _lib.lib()
        s = ops._stream()
        hp, n_hist, npn, nip, n, hs, ns = self._draw_args(hist, new, new_idx)
        iters = hist.shape[0]
        # ---- boxes
        if self.fixed_boxes is not None:
            boxes = np.asarray(self.fixed_boxes, np.int32).reshape(self.world, 6)
        else:
            n_out = min(self.KD_SAMPLES, n)
            stride = max(1, n // max(n_out, 1))
            n_out = min(n_out, (n + stride - 1) // stride)
            if self._cells is None or self._cells.shape[0] < n_out:
                self._cells = torch.empty((self.KD_SAMPLES, 3), dtype=torch.int32, device=self.device)
                self._cells_host = torch.empty((self.KD_SAMPLES, 3), dtype=torch.int32).pin_memory()
            check(L.pin_dp_sample_cells(pool_coord.data_ptr(), hp, n_hist, npn, nip, n, stride, n_out,
                                        float(np.float32(resolution)), self._cells.data_ptr(), s), "pin_dp_sample_cells")
            self._cells_host[:n_out].copy_(self._cells[:n_out], non_blocking=True)
            torch.cuda.current_stream().synchronize()
            boxes = kd_boxes(self._cells_host[:n_out].numpy(), self.world)
        self.boxes = boxes
        if self.world > 1 and getattr(self.comm, "kind", "").startswith("none"):
            self._boxes_host.copy_(torch.from_numpy(boxes))  # (a rank emulated alone: nobody to agree with)
            self.boxes_dev.copy_(self._boxes_host, non_blocking=True)
        else:
            # rank 0's boxes are everybody's: exact fp32 halves through the all-reduce (pin_dp_boxes_decode)
            if self._halves is None:
                self._halves = torch.zeros((self.world * 12,), dtype=torch.float32, device=self.device)
                self._halves_host = torch.zeros((self.world * 12,), dtype=torch.float32).pin_memory()
            hh = self._halves_host.numpy().reshape(-1, 2)
            if self.rank == 0:
                v = boxes.astype(np.int64).reshape(-1)
                hh[:, 0], hh[:, 1] = (v >> 16).astype(np.float32), (v & 0xFFFF).astype(np.float32)
            else:
                hh[:] = 0.0
            self._halves.copy_(self._halves_host, non_blocking=True)
            if self.world > 1:
                self.comm.allreduce(self._halves, self._halves)
            check(L.pin_dp_boxes_decode(self._halves.data_ptr(), self.world, self.boxes_dev.data_ptr(), s), "pin_dp_boxes_decode")
        rg = self.regions(resolution, reach)
        # ---- halo of the feature rows
        rows = pos.shape[0]
        if self.halo_rows is None or self.halo_rows.shape[0] < rows:
            cap_rows = int(rows * 1.25) + 1024
            self.halo_rows = torch.empty((cap_rows,), dtype=torch.int32, device=self.device)
            self.owner = torch.empty((cap_rows,), dtype=torch.uint8, device=self.device)
        need = int(L.pin_maint_workspace_bytes(rows))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty((int(need * 1.25) + 1024,), dtype=torch.uint8, device=self.device)
        check(L.pin_dp_mark_halo(C.byref(rg), pos.data_ptr(), rows, self.halo_rows.data_ptr(), self.halo_rows.shape[0],
                                 self._cnt.data_ptr(), self.owner.data_ptr(), None if lazy_pending is None else lazy_pending.data_ptr(),
                                 self._ws.data_ptr(), self._ws.numel(), s), "pin_dp_mark_halo")
        # ---- the private rows of every box, box after box (what each rank publishes at the end of the call)
        if self.lists is None or self.lists.shape[0] < rows:
            self.lists = torch.empty((int(rows * 1.25) + 1024,), dtype=torch.int32, device=self.device)
            self.offsets = torch.zeros((self.world + 1,), dtype=torch.int32, device=self.device)
        need = int(L.pin_dp_owner_lists_workspace_bytes(rows, self.world))
        if self._ws.numel() < need:
            self._ws = torch.empty((int(need * 1.25) + 1024,), dtype=torch.uint8, device=self.device)
        check(L.pin_dp_owner_lists(self.owner.data_ptr(), rows, self.world, self.lists.data_ptr(), self.offsets.data_ptr(),
                                   self._ws.data_ptr(), self._ws.numel(), s), "pin_dp_owner_lists")
        # ---- partition of every drawn batch
        W1 = self.world + 1
        if self.counts is None or self.counts.shape[0] < iters:
            self.counts = torch.zeros((max(iters, 16), 2), dtype=torch.int32, device=self.device)
            self.counts_host = torch.zeros((max(iters, 16) * 2 + 2 + W1,), dtype=torch.int32).pin_memory()
            self._own_cnt = torch.zeros((1,), dtype=torch.int32, device=self.device)
        want_cap = int(n / self.world * 1.15) + 1024
        want_ecap = (int((n + decimation - 1) // decimation / self.world * 1.25) + 256) if eikonal else 0
        while True:
            if self.sel is None or self.cap < want_cap or self.ecap < want_ecap or self.sel.shape[0] < iters * self.cap:
                self.cap, self.ecap = max(self.cap, want_cap), max(self.ecap, want_ecap)
                self.sel = torch.empty((max(iters, 1) * self.cap,), dtype=torch.int32, device=self.device)
                self.esel = torch.empty((max(iters, 1) * max(self.ecap, 1),), dtype=torch.int32, device=self.device)
            ecap = self.ecap if eikonal else 0
            pool_rows = pool_coord.shape[0] if pool_rows is None else int(pool_rows)
            if self._pool_region is None or self._pool_region.shape[0] < pool_rows:
                self._pool_region = torch.empty((int(pool_rows * 1.25) + 1024,), dtype=torch.uint8, device=self.device)
            want_surf = color_pending is not None and pool_label is not None
            if want_surf and (self.surf_counts is None or self.surf_counts.shape[0] < iters):
                self.surf_counts = torch.zeros((max(iters, 16),), dtype=torch.int32, device=self.device)
            check(L.pin_dp_partition(C.byref(rg), pool_coord.data_ptr(), hp, n_hist, npn, nip, n, int(decimation), iters, hs, ns,
                                     self.sel.data_ptr(), self.cap, self.esel.data_ptr(), ecap, self.counts.data_ptr(), pool_rows,
                                     self._pool_region.data_ptr(), pool_label.data_ptr() if want_surf else None, float(surface_range),
                                     self.surf_counts.data_ptr() if want_surf else None, s), "pin_dp_partition")
            if reuse_records:
                # this rank's pool samples, compacted in pool order: ONE neighbour search over them serves every iteration of the
                # call (engine.MapTrainer.run_shards); their number comes back with the partition's counts
                want_own = int(pool_rows / self.world * 1.3) + 4096
                if self.own_coord is None or self.own_coord.shape[0] < want_own or self.pool_to_own.shape[0] < pool_rows:
                    self.own_cap = max(self.own_cap, want_own)
                    self.own_coord = torch.empty((self.own_cap, 3), dtype=torch.float32, device=self.device)
                    self.pool_to_own = torch.empty((int(pool_rows * 1.25) + 1024,), dtype=torch.int32, device=self.device)
                need = 4 * (pool_rows // 256 + 2) + 512
                if self._ws.numel() < need:
                    self._ws = torch.empty((int(need * 1.25) + 1024,), dtype=torch.uint8, device=self.device)
                check(L.pin_dp_own_pool(self._pool_region.data_ptr(), pool_rows, self.rank, pool_coord.data_ptr(), self.own_coord.data_ptr(),
                                        self.own_coord.shape[0], self.pool_to_own.data_ptr(), self._own_cnt.data_ptr(), self._ws.data_ptr(),
                                        self._ws.numel(), s), "pin_dp_own_pool")
            ch = self.counts_host
            ch[:2 * iters].copy_(self.counts[:iters].reshape(-1), non_blocking=True)
            ch[2 * iters:2 * iters + 1].copy_(self._cnt, non_blocking=True)
            ch[2 * iters + 1:2 * iters + 1 + W1].copy_(self.offsets, non_blocking=True)
            if reuse_records:
                ch[2 * iters + 1 + W1:2 * iters + 2 + W1].copy_(self._own_cnt, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            cnt = ch[:2 * iters].numpy().reshape(iters, 2).copy()
            n_own = int(ch[2 * iters + 1 + W1]) if reuse_records else None
            if n_own is not None and n_own > self.own_coord.shape[0]:
                self.own_cap = int(n_own * 1.1) + 4096  # a clustered pool: more of it in this box than its share
                self.own_coord = None
                continue
            if cnt[:, 0].max(initial=0) <= self.cap and (not eikonal or cnt[:, 1].max(initial=0) <= self.ecap):
                break
            # an unbalanced cut (a clustered pool): grow the lists and partition again
            want_cap = max(want_cap, int(cnt[:, 0].max() * 1.1) + 1024)
            want_ecap = max(want_ecap, int(cnt[:, 1].max() * 1.1) + 256) if eikonal else 0
        self.n_own = n_own
        self.pool_rows = pool_rows
        self.n_main, self.n_eik = cnt[:, 0].astype(int), (cnt[:, 1].astype(int) if eikonal else np.zeros(iters, int))
        self.eik_cap = ecap
        self.n_halo = int(ch[2 * iters])
        self.offsets_host = ch[2 * iters + 1:2 * iters + 1 + W1].numpy().astype(np.int64).copy()
        self._check_replicas(hist, iters, n, decimation, eikonal, rows, pool_rows)
        self.seg_rows = int(np.diff(self.offsets_host).max(initial=0))
        if self.n_halo > self.halo_rows.shape[0]:
            raise RuntimeError("halo list overflow")  # (cannot happen: the list is sized for every row)
        # ---- exchange buffer [decoder grads | halo-row grads (| the colour table's halo-row grads)] and the halo's compact
        # Adam moments (geometry first, colour behind)
        self.tables = 2 if color_pending is not None else 1
        if color_pending is not None:
            check(L.pin_dp_exclude_rows(self.halo_rows.data_ptr(), self.n_halo, color_pending.data_ptr(), s), "pin_dp_exclude_rows")
        nx = nd + 8 * self.n_halo * self.tables
        if self.xbuf is None or self.xbuf.numel() < nx or self._nd != nd:
            capx = nd + int(8 * self.n_halo * self.tables * 1.25) + 1024
            self.xbuf = torch.zeros((capx,), dtype=torch.float32, device=self.device)
            self.hm = torch.zeros((capx,), dtype=torch.float32, device=self.device)
            self.hv = torch.zeros((capx,), dtype=torch.float32, device=self.device)
            self._nd = nd
        else:
            self.hm[:8 * self.n_halo * self.tables].zero_()
            self.hv[:8 * self.n_halo * self.tables].zero_()
        self.nd = nd
        self.stats = dict(rows=rows, halo_rows=self.n_halo, halo_fraction=self.n_halo / max(rows, 1),
                          exchange_bytes=4 * nx, merge=self.merge,
                          merge_bytes_per_call=((40 + 32 * (self.tables - 1)) * self.seg_rows * self.world + 12 * self.n_halo)
                          if self.merge == "gather" else (32 * self.tables + 8) * rows, samples_min=int(self.n_main.min()) if iters else 0, samples_max=int(self.n_main.max()) if iters else 0,
                          samples_ideal=n / self.world)
        self._hist, self._new, self._new_idx, self._pool_coord = hist, new, new_idx, pool_coord
        return self

    def _check_replicas(self, hist, iters, n, decimation, eikonal, rows, pool_rows):
        """Only the boxes are agreed through an exchange; the halo list, the owner lists and the partition are derived by every
        rank from ITS replica of the map and the pool, and publish() indexes those local lists with the other ranks' payloads.
        Replicas that differ in one bit would write rows to the wrong places or hang the all-gather on mismatched counts, so
        every call compares a signature (pin_dp_signature: halo count + checksum, owner-list offsets, checksum of the first drawn
        batch; host: rows, pool size, batch) through one tiny all-reduce and raises when the ranks disagree; the per-rank
        sample counts ride along and must add up to the batch."""
        if self.world <= 1 or getattr(self.comm, "kind", "").startswith("none"):
            return
        W, L = self.world, _lib.lib()
        n_first = min(hist.shape[1], 1 << 16)
        words = W + 4 + 2 * iters
        host_words = [rows & 0xFFFFFFFF, pool_rows & 0xFFFFFFFF, n & 0xFFFFFFFF, iters, int(decimation), int(bool(eikonal))]
        nf = 2 * (words + len(host_words))
        if getattr(self, "_sig", None) is None or self._sig[0].numel() < nf:
            self._sig = (torch.zeros((nf + 64,), dtype=torch.float32, device=self.device),
                         torch.zeros((nf + 64,), dtype=torch.float32, device=self.device),
                         torch.zeros((2 * (nf + 64),), dtype=torch.float32).pin_memory())
        mine, total, host = self._sig
        hw = torch.tensor([[float(v >> 16), float(v & 0xFFFF)] for v in host_words], dtype=torch.float32).reshape(-1)
        mine[2 * words:nf].copy_(hw, non_blocking=True)
        check(L.pin_dp_signature(self.halo_rows.data_ptr(), self._cnt.data_ptr(), self.offsets.data_ptr(), W, hist.data_ptr(),
                                 n_first, self.counts.data_ptr(), 2 * iters, mine.data_ptr(), ops._stream()), "pin_dp_signature")
        self.comm.allreduce(mine[:nf], total[:nf])
        host[:nf].copy_(mine[:nf], non_blocking=True)
        host[nf:2 * nf].copy_(total[:nf], non_blocking=True)
        torch.cuda.current_stream().synchronize()
        h = host[:2 * nf].numpy().astype(np.int64)
        own = (h[0:nf:2] << 16) + h[1:nf:2]
        tot = (h[nf:2 * nf:2] << 16) + h[nf + 1:2 * nf:2]
        names = ["halo rows"] + [f"owner-list offset {i}" for i in range(W + 1)] + ["halo-list checksum", "first-batch checksum"]
        names += [f"count[{i // 2}][{i % 2}]" for i in range(2 * iters)]
        names += ["feature rows", "pool rows", "batch", "iterations", "decimation", "eikonal"]
        same = list(range(W + 4)) + list(range(words, words + len(host_words)))
        bad = [names[i] for i in same if tot[i] != W * own[i]]
        if bad:
            raise RuntimeError(f"spatially sharded mapper, rank {self.rank}: the ranks' replicas of the map / sample pool have diverged "
                               f"({', '.join(bad)} differ between ranks); refusing to exchange rows over mismatched lists")
        cnt = tot[W + 4:words].reshape(iters, 2)
        n_eik = (n + decimation - 1) // decimation if eikonal else 0
        if (cnt[:, 0] != n).any() or (eikonal and (cnt[:, 1] != n_eik).any()):
            raise RuntimeError(f"spatially sharded mapper: the ranks' shares of a batch do not add up to the batch "
                               f"(main {cnt[:, 0].tolist()} of {n}, Eikonal {cnt[:, 1].tolist()} of {n_eik})")

    # ------------------------------------------------------------------ per group of iterations
    def gather(self, pool: dict, global_coord: bool, C_color: int, it0: int, gn: int, out: dict, query_all: torch.Tensor, eps: float):
        """Mapper.get_batch of iterations it0 .. it0 + gn - 1 for this rank's samples (pin_dp_gather): out = dict of
        [gn][cap] buffers (coord, label, weight, ts, color), query_all [gn][cap + 6 ecap][3]."""
        hist, new = self._hist, self._new
        n_hist = hist.shape[1]
        n_new = 0 if new is None else new.shape[1]
        check(_lib.lib().pin_dp_gather(
            (pool["global_coord"] if global_coord else pool["coord"]).data_ptr(), pool["sdf_label"].data_ptr(), pool["weight"].data_ptr(),
            pool["ts"].data_ptr(), pool["color"].data_ptr() if C_color else None, C_color,
            hist.data_ptr() + 8 * it0 * n_hist, n_hist, None if new is None else new.data_ptr() + 8 * it0 * n_new,
            None if new is None else self._new_idx.data_ptr(), n_hist, n_new,
            self.sel.data_ptr() + 4 * it0 * self.cap, self.cap, self.esel.data_ptr() + 4 * it0 * max(self.eik_cap, 0), self.eik_cap,
            self.counts.data_ptr() + 8 * it0, gn, out["coord"].data_ptr(), out["label"].data_ptr(), out["weight"].data_ptr(),
            out["ts"].data_ptr(), None if out.get("color") is None else out["color"].data_ptr(), query_all.data_ptr(),
            float(np.float32(eps)), ops._stream()), "pin_dp_gather")

    def records(self, k: int):
        """(rec_nbr [n_own][k][4], rec_nn [n_own]) buffers for the one search over this rank's pool samples."""
        if self.rec is None or self.rec[0].shape[0] < self.n_own or self.rec[0].shape[1] != k:
            cap = max(self.own_coord.shape[0], self.n_own)
            self.rec = (torch.empty((cap, k, 4), dtype=torch.float32, device=self.device), torch.empty((cap,), dtype=torch.int32, device=self.device))
        return self.rec[0][:self.n_own], self.rec[1][:self.n_own]

    def gather_records(self, it0: int, gn: int, k: int, nbr_all: torch.Tensor, nn_all: torch.Tensor):
        """The records of this rank's samples of iterations it0 .. it0 + gn - 1 into the trainer's record buffers
        (pin_dp_gather_records); the Eikonal probes behind them are searched per iteration."""
        hist, new = self._hist, self._new
        n_hist = hist.shape[1]
        n_new = 0 if new is None else new.shape[1]
        check(_lib.lib().pin_dp_gather_records(
            self.rec[0].data_ptr(), self.rec[1].data_ptr(), k, self.pool_to_own.data_ptr(), hist.data_ptr() + 8 * it0 * n_hist, n_hist,
            None if new is None else new.data_ptr() + 8 * it0 * n_new, None if new is None else self._new_idx.data_ptr(), n_hist, n_new,
            self.sel.data_ptr() + 4 * it0 * self.cap, self.cap, self.eik_cap, self.counts.data_ptr() + 8 * it0, gn, nbr_all.data_ptr(),
            nn_all.data_ptr(), ops._stream()), "pin_dp_gather_records")

    # ------------------------------------------------------------------ per iteration
    def exchange(self, feats: torch.Tensor, gfeat: torch.Tensor, step: int, coef: torch.Tensor, t_max: int, b1, b2, eps,
                 on_allreduce=None, color=None):
        """After the backward pass of iteration `step`: halo gradients into the exchange buffer (its head already holds
        the decoder gradients), ONE all-reduce, the dense Adam step on the halo rows -- all on the caller's stream.
        color = (colour features, their gradient table) of a colour map: the same rows of the second table ride in the same
        message.  engine.MapTrainer runs the three parts apart (pack_halo / allreduce_range / halo_step) to overlap the
        all-reduce with work that does not depend on it."""
        self.pack_halo(gfeat, color)
        self.allreduce_range(0, self.nd + 8 * self.n_halo * self.tables, on_allreduce)
        self.halo_step(feats, step, coef, t_max, b1, b2, eps, color)

    def pack_halo(self, gfeat: torch.Tensor, color=None):
        """Halo-row gradients of the feature table(s) -> the compact buffer behind the decoder gradients (rows cleared)."""
        L, s, nh = _lib.lib(), ops._stream(), self.n_halo
        check(L.pin_dp_halo_pack(self.halo_rows.data_ptr(), nh, gfeat.data_ptr(), self.xbuf.data_ptr() + 4 * self.nd, s),
              "pin_dp_halo_pack")
        if color is not None:
            check(L.pin_dp_halo_pack(self.halo_rows.data_ptr(), nh, color[1].data_ptr(), self.xbuf.data_ptr() + 4 * (self.nd + 8 * nh), s),
                  "pin_dp_halo_pack")

    def allreduce_range(self, lo: int, hi: int, on_allreduce=None):
        """In-place SUM all-reduce of xbuf[lo:hi] on the current stream ([0, nd) = decoder gradients, [nd, nx) = halo rows)."""
        if hi <= lo:
            return
        if on_allreduce is not None:
            on_allreduce(True)
        self.comm.allreduce(self.xbuf[lo:hi], self.xbuf[lo:hi])
        if on_allreduce is not None:
            on_allreduce(False)

    def halo_step(self, feats: torch.Tensor, step: int, coef: torch.Tensor, t_max: int, b1, b2, eps, color=None):
        """The same dense Adam step on the halo rows of every rank, from the summed gradients."""
        L, s, nh = _lib.lib(), ops._stream(), self.n_halo
        check(L.pin_dp_halo_adam(self.halo_rows.data_ptr(), nh, feats.data_ptr(), self.xbuf.data_ptr() + 4 * self.nd,
                                 self.hm.data_ptr(), self.hv.data_ptr(), int(step), coef.data_ptr(), int(t_max), float(b1), float(b2),
                                 float(eps), s), "pin_dp_halo_adam")
        if color is not None:
            check(L.pin_dp_halo_adam(self.halo_rows.data_ptr(), nh, color[0].data_ptr(), self.xbuf.data_ptr() + 4 * (self.nd + 8 * nh),
                                     self.hm.data_ptr() + 32 * nh, self.hv.data_ptr() + 32 * nh, int(step), coef.data_ptr(), int(t_max),
                                     float(b1), float(b2), float(eps), s), "pin_dp_halo_adam")

    # ------------------------------------------------------------------ end of the call
    def publish(self, feats: torch.Tensor, scratch: torch.Tensor, certainty: torch.Tensor, certainty0: torch.Tensor,
                cert_scratch: torch.Tensor, ts_update: torch.Tensor, color=None):
        """The trained table, the certainties and the timestamps on every rank (rows [0, n_rows); the padding row behind
        them never trains).  merge "gather": every rank packs the (features, certainty, ts_update) of the rows it OWNS
        (a private row is only ever touched by its owner's samples, so the owner's values are the merged values), one
        all-gather, everybody unpacks the others' rows; the halo rows' features are identical everywhere already, their
        certainty (sum of the ranks' deltas) and ts_update (max) go through the side-effect exchange in compact form.
        merge "reduce": owner ? row : 0 all-reduced into the table + the side-effect exchange over every row.
        scratch: n_rows * 8 floats that may be overwritten (the lazy optimiser's moment array: rebuilt by the next call)."""
        L, s = _lib.lib(), ops._stream()
        rows, W = self.stats["rows"], self.world
        if self.merge != "gather":
            check(L.pin_dp_owner_pack(self.owner.data_ptr(), self.rank, feats.data_ptr(), rows, scratch.data_ptr(), s), "pin_dp_owner_pack")
            self.comm.allreduce(scratch[:8 * rows], feats.reshape(-1)[:8 * rows])
            if color is not None:  # color = (colour features, a scratch table of their size)
                check(L.pin_dp_owner_pack(self.owner.data_ptr(), self.rank, color[0].data_ptr(), rows, color[1].data_ptr(), s), "pin_dp_owner_pack")
                self.comm.allreduce(color[1][:8 * rows], color[0].reshape(-1)[:8 * rows])
            self.comm.sync_side_effects(certainty, certainty0, cert_scratch, ts_update)
            return
        seg, nh = self.seg_rows, self.n_halo
        rec = 18 if color is not None else 10
        cf = None if color is None else color[0].data_ptr()
        if self._send is None or self._send.numel() < rec * seg or self._recv.numel() < rec * seg * W:
            cap = int(rec * seg * 1.1) + 1024
            self._send = torch.empty((cap,), dtype=torch.float32, device=self.device)
            self._recv = torch.empty((cap * W,), dtype=torch.float32, device=self.device)
        if self._side is None or self._side[0].numel() < nh:
            cap = int(nh * 1.25) + 1024
            self._side = (torch.empty((cap,), dtype=torch.float32, device=self.device), torch.empty((cap,), dtype=torch.float32, device=self.device),
                          torch.empty((cap,), dtype=torch.float32, device=self.device), torch.empty((cap,), dtype=torch.int32, device=self.device))
        mine = int(self.offsets_host[self.rank + 1] - self.offsets_host[self.rank])
        check(L.pin_dp_rows_pack(self.lists.data_ptr() + 4 * int(self.offsets_host[self.rank]), mine, feats.data_ptr(), certainty.data_ptr(),
                                 ts_update.data_ptr(), cf, self._send.data_ptr(), s), "pin_dp_rows_pack")
        self.comm.allgather(self._send[:rec * seg], self._recv[:rec * seg * W])
        check(L.pin_dp_rows_unpack(self._recv.data_ptr(), seg, self.lists.data_ptr(), self.offsets.data_ptr(), W, self.rank,
                                   feats.data_ptr(), certainty.data_ptr(), ts_update.data_ptr(), cf, s), "pin_dp_rows_unpack")
        if nh:
            cc, cc0, csc, cts = (t[:nh] for t in self._side)
            check(L.pin_dp_halo_side_gather(self.halo_rows.data_ptr(), nh, certainty.data_ptr(), certainty0.data_ptr(), ts_update.data_ptr(),
                                            cc.data_ptr(), cc0.data_ptr(), cts.data_ptr(), s), "pin_dp_halo_side_gather")
            self.comm.sync_side_effects(cc, cc0, csc, cts)
            check(L.pin_dp_halo_side_scatter(self.halo_rows.data_ptr(), nh, cc.data_ptr(), cts.data_ptr(), certainty.data_ptr(),
                                             ts_update.data_ptr(), s), "pin_dp_halo_side_scatter")
