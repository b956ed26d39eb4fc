"""Device-resident sample pool of the mapper (Mapper.process_frame data path,
utils/mapper.py:162-449): capacity-managed, double-buffered arrays that the sampling kernel
appends to in place and the filter kernel compacts from one buffer into the other -- no
per-frame torch.cat / boolean-mask copies of the whole pool.

Random numbers are drawn with torch in the reference's order (randn surface, rand front, rand
behind, randint discard), so a seeded run consumes the generator exactly as the reference."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _lib, ops
from ._lib import PoolArrays, SampleParams, check

FIELDS = ("coord", "global_coord", "sdf_label", "weight", "ts", "color", "sem_label")


def sample_params(cfg, pose: np.ndarray, frame_id: int) -> SampleParams:
    """pin_sample_params from a (reference or PinConfig) config object."""
    p = SampleParams()
    p.surface_n, p.front_n, p.behind_n = int(cfg.surface_sample_n), int(cfg.free_front_n), int(cfg.free_behind_n)
    p.dist_weight_on, p.behind_dropoff_on = int(bool(cfg.dist_weight_on)), int(bool(cfg.behind_dropoff_on))
    p.frame_id = int(frame_id)
    p.surface_range = float(cfg.surface_sample_range_m)
    p.free_begin_ratio, p.free_end_dist = float(cfg.free_sample_begin_ratio), float(cfg.free_sample_end_dist_m)
    p.dist_weight_scale, p.max_range = float(cfg.dist_weight_scale), float(cfg.max_range)
    T = np.asarray(pose, np.float64)
    for i in range(12):
        p.pose[i] = float(T[i // 4, i % 4])
    return p


class SamplePool:
    def __init__(self, device="cuda", color_channels: int = 0, capacity: int = 1 << 16, semantic: bool = False):
        self.device = torch.device(device)
        self.C = int(color_channels)
        self.semantic = bool(semantic)  # carry sem_label_pool (int32 per sample)
        self.n = 0          # samples in the pool
        self.n_cur = 0      # of which belong to the newest frame (the tail)
        self.cap = 0
        self.bufs = [None, None]  # two generations: compaction goes from bufs[0] to bufs[1], then they swap
        self._ensure(capacity)
        # [0] kept by the window / the compaction, [1] kept among the newest frame's samples, [2] a caller's count that is read
        # back together with [0] (Mapper.process_frame: the surface-point count) -- one block, one read-back, no torch.stack
        self.counts = torch.zeros((4,), dtype=torch.int32, device=self.device)
        self.counts_host = torch.zeros((4,), dtype=torch.int32).pin_memory()
        self.mask = self.true_index = self.ws = None

    # ------------------------------------------------------------------ storage
    def _alloc(self, cap):
        d = self.device
        b = dict(coord=torch.empty((cap, 3), dtype=torch.float32, device=d),
                 global_coord=torch.empty((cap, 3), dtype=torch.float32, device=d),
                 sdf_label=torch.empty((cap,), dtype=torch.float32, device=d),
                 weight=torch.empty((cap,), dtype=torch.float32, device=d),
                 ts=torch.empty((cap,), dtype=torch.int32, device=d),
                 color=torch.empty((cap, self.C), dtype=torch.float32, device=d) if self.C else None,
                 sem_label=torch.empty((cap,), dtype=torch.int32, device=d) if self.semantic else None)
        return b

    def _ensure(self, need):
        if need <= self.cap:
            return
        cap = max(int(need), int(self.cap * 1.5), 1 << 16)
        new = self._alloc(cap)
        old = self.bufs[0]
        if old is not None and self.n:
            for k in FIELDS:
                if new[k] is not None:
                    new[k][:self.n].copy_(old[k][:self.n])
        self.bufs = [new, None]  # the second generation is (re)allocated lazily by filter()
        self.cap = cap

    def _arrays(self, b, offset=0) -> PoolArrays:
        a = PoolArrays()
        a.coord, a.global_coord = b["coord"][offset:].data_ptr(), b["global_coord"][offset:].data_ptr()
        a.sdf_label, a.weight, a.ts = b["sdf_label"][offset:].data_ptr(), b["weight"][offset:].data_ptr(), b["ts"][offset:].data_ptr()
        a.color = b["color"][offset:].data_ptr() if self.C else None
        a.color_channels = self.C
        a.sem_label = b["sem_label"][offset:].data_ptr() if self.semantic else None
        return a

    def view(self, name) -> Optional[torch.Tensor]:
        t = self.bufs[0][name]
        return None if t is None else t[:self.n]

    def clear(self):
        self.n = self.n_cur = 0

    def adopt(self, **tensors):
        """Replace the content by externally built tensors (e.g. after transform_data_pool)."""
        n = tensors["sdf_label"].shape[0]
        self.n = 0
        self._ensure(n)
        for k, t in tensors.items():
            if t is not None and self.bufs[0][k] is not None:
                self.bufs[0][k][:n].copy_(t.to(self.bufs[0][k].dtype))
        self.n = n
        self.n_cur = min(self.n_cur, n)

    # ------------------------------------------------------------------ K12
    def append_samples(self, scan: torch.Tensor, sp: SampleParams, rnd=None, sem_labels: Optional[torch.Tensor] = None) -> int:
        """DataSampler.sample + pool append for one frame.  scan [N, 3(+C)] float32 rows in the
        sensor frame (colour channels after xyz when the pool carries colour).  rnd = optional
        (surface, front, behind) draws; by default they are drawn here with torch.randn / rand in
        the reference's order and shapes."""
        if not (scan.is_cuda and scan.dtype == torch.float32 and scan.dim() == 2 and scan.stride(1) == 1):
            raise RuntimeError("scan must be a float32 device tensor with unit column stride")
        N, stride = scan.shape[0], scan.stride(0)
        if self.C and scan.shape[1] < 3 + self.C:
            raise RuntimeError("scan rows carry no colour channels")
        A = 1 + sp.surface_n + sp.front_n + sp.behind_n
        if rnd is None:
            dev = scan.device
            rnd = (torch.randn(N * sp.surface_n, 1, device=dev), torch.rand(N * sp.front_n, 1, device=dev),
                   torch.rand(N * sp.behind_n, 1, device=dev))
        self._ensure(self.n + N * A)
        if self.semantic:  # per-point labels of this scan (frame_label_torch); None: every sample gets label 0
            if sem_labels is not None:
                if sem_labels.shape[0] != N:
                    raise RuntimeError("one semantic label per scan point is needed")
                sem_labels = sem_labels.detach().to(device=scan.device, dtype=torch.int32).contiguous()
                sp.sem_labels = sem_labels.data_ptr()
            else:
                sp.sem_labels = None
        elif sem_labels is not None:
            raise RuntimeError("this pool carries no semantic labels (SamplePool(semantic=True))")
        out = self._arrays(self.bufs[0], self.n)
        colors = scan.data_ptr() + 12 if self.C else None
        check(_lib.lib().pin_sample_rays(C.byref(sp), scan.data_ptr(), colors, stride, N, ops._ptr(rnd[0].reshape(-1)),
                                         ops._ptr(rnd[1].reshape(-1)), ops._ptr(rnd[2].reshape(-1)), C.byref(out),
                                         ops._stream()), "pin_sample_rays")
        self.n += N * A
        self.n_cur = N * A
        return N * A

    # ------------------------------------------------------------------ K13
    def filter(self, origin, radius: float, capacity: int, discard_index: Optional[torch.Tensor] = None):
        """Distance window + random discard + ordered compaction (utils/mapper.py:303-346).
        Returns (pool_sample_count, cur_sample_count)."""
        self.filter_begin(origin, radius, capacity)
        return self.filter_finish(capacity, discard_index)

    def filter_begin(self, origin, radius: float, capacity: int):
        """Window mask (+ list of kept indices when the capacity can be exceeded): launch only.  The kept
        count lands in `self.counts[0]`; callers that have other read-backs pending fetch them together."""
        L = _lib.lib()
        n, dev = self.n, self.device
        self._over = False
        if n == 0:
            return
        if self.mask is None or self.mask.numel() < self.cap:
            self.mask = torch.empty((self.cap,), dtype=torch.uint8, device=dev)
            self.true_index = None
            self.ws = ops.pool_workspace(self.cap, dev)
        if self.bufs[1] is None or self.bufs[1]["sdf_label"].shape[0] < self.cap:
            self.bufs[1] = self._alloc(self.cap)
        o = np.ascontiguousarray(np.asarray(origin, dtype=np.float64))
        over = n > capacity  # only then can the window keep more than `capacity` samples
        if over and (self.true_index is None or self.true_index.numel() < self.cap):
            self.true_index = torch.empty((self.cap,), dtype=torch.int32, device=dev)
        self._over = over
        check(L.pin_pool_window_mask(self.bufs[0]["global_coord"].data_ptr(), n, o.ctypes.data, float(radius),
                                     self.mask.data_ptr(), self.true_index.data_ptr() if over else None,
                                     self.counts.data_ptr(), self.ws.data_ptr(), self.ws.numel(), ops._stream()),
              "pin_pool_window_mask")

    def filter_finish(self, capacity: int, discard_index: Optional[torch.Tensor] = None, kept: Optional[int] = None,
                      before_sync=None):
        """Random discard (the reference's torch.randint draw) + compaction + the two counts (one read-back;
        one more for `kept` if the caller has not fetched it).  before_sync(): called once the launches are queued, in
        front of the read-back (the caller queues work for other streams there, to run while this one waits)."""
        L = _lib.lib()
        n, dev = self.n, self.device
        if n == 0:
            return 0, 0
        stream = ops._stream()
        src = self.bufs[0]
        if self._over:
            if kept is None:
                kept = int(self.counts[0].item())
            if kept > capacity:
                nd = kept - capacity
                if discard_index is None:
                    discard_index = torch.randint(0, kept, (nd,), device=dev)
                if discard_index.numel() != nd:
                    raise RuntimeError("discard_index must hold kept - pool_capacity draws")
                check(L.pin_pool_discard(self.mask.data_ptr(), self.true_index.data_ptr(),
                                         ops._ptr(discard_index, torch.int64), nd, stream), "pin_pool_discard")
        a, b = self._arrays(src), self._arrays(self.bufs[1])
        check(L.pin_pool_compact(C.byref(a), C.byref(b), self.mask.data_ptr(), n, self.n_cur, self.counts.data_ptr(),
                                 self.ws.data_ptr(), self.ws.numel(), stream), "pin_pool_compact")
        self.counts_host.copy_(self.counts, non_blocking=True)
        if before_sync is not None:
            before_sync()
        torch.cuda.current_stream().synchronize()
        self.bufs = [self.bufs[1], self.bufs[0]]
        self.n, self.n_cur = int(self.counts_host[0]), int(self.counts_host[1])
        return self.n, self.n_cur

    def read_counts(self, n: int = 3):
        """counts[:n] on the host (one copy into the pinned block + one stream synchronisation)."""
        self.counts_host[:n].copy_(self.counts[:n], non_blocking=True)
        torch.cuda.current_stream().synchronize()
        return self.counts_host[:n].tolist()
