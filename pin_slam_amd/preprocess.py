"""SLAMDataset.preprocess_frame data path (dataset/slam_dataset.py:359-505) on libpinhip:
function-level mirrors of the reference helpers it calls -- ``voxel_down_sample_torch``
(utils/tools.py:583), ``crop_frame`` (slam_dataset.py:1229), ``intrinsic_correct`` (:1251),
``deskewing`` (utils/tools.py:747) -- with the reference's signatures, and :class:`ScanPreprocessor`
that chains them for one frame.  ``patch_reference()`` rebinds the reference's module-level
names to these functions (drop-in mode)."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import os

import numpy as np
import torch

from . import _lib, ops
from ._lib import check

_ws_cache = {}


def _ws(nbytes: int, device) -> torch.Tensor:
    key = (str(device), ops._stream())  # (one scratch per stream: launches of two streams never share it)
    t = _ws_cache.get(key)
    if t is None or t.numel() < nbytes:
        t = torch.empty((int(nbytes * 1.25) + 256,), dtype=torch.uint8, device=device)
        _ws_cache[key] = t
    return t


def _dev_f32(t: torch.Tensor) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError("libpinhip needs device (HIP) tensors; there is no CPU path")
    return t.detach().to(torch.float32)


def voxel_down_sample_torch(points: torch.Tensor, voxel_size: float) -> torch.Tensor:
    """Index (int64, ordered by voxel id) of the point closest to its voxel centre, per voxel."""
    return _voxel_down_sample_i32(points, voxel_size).long()


def _voxel_down_sample_i32(points: torch.Tensor, voxel_size: float) -> torch.Tensor:
    """The same as the int32 tensor the kernel wrote (ScanPreprocessor feeds it straight to the row gather)."""
    L = _lib.lib()
    pts = _dev_f32(points).contiguous()
    n = pts.shape[0]
    ws = _ws(L.pin_maint_workspace_bytes(n), pts.device)
    sel = torch.empty((n,), dtype=torch.int32, device=pts.device)
    cnt = torch.empty((1,), dtype=torch.int32, device=pts.device)
    args = (pts.data_ptr(), n, float(np.float32(voxel_size)), sel.data_ptr(), cnt.data_ptr(), ws.data_ptr(), ws.numel(), ops._stream())
    check(L.pin_voxel_downsample_fast(*args), "pin_voxel_downsample_fast")
    c = int(cnt.item())
    if c < 0:  # voxel ids too wide for the one-word sort key (extent > ~5 000 voxels per axis): the general form
        check(L.pin_voxel_downsample(*args), "pin_voxel_downsample")
        c = int(cnt.item())
    return sel[:c]


def crop_frame(points: torch.Tensor, ts: Optional[torch.Tensor], min_z_th=-3.0, max_z_th=100.0, min_range=2.75,
               max_range=100.0):
    L = _lib.lib()
    pts = _dev_f32(points).contiguous()
    n, w = pts.shape
    t32 = None if ts is None else _dev_f32(ts).reshape(-1).contiguous()
    out = torch.empty_like(pts)
    ts_out = None if ts is None else torch.empty_like(t32)
    cnt = torch.empty((1,), dtype=torch.int32, device=pts.device)
    ws = _ws(L.pin_pool_workspace_bytes(n) + n, pts.device)
    check(L.pin_crop_frame(pts.data_ptr(), w, n, None if t32 is None else t32.data_ptr(), float(min_z_th), float(max_z_th),
                           float(min_range), float(max_range), out.data_ptr(), None if ts_out is None else ts_out.data_ptr(),
                           cnt.data_ptr(), ws.data_ptr(), ws.numel(), ops._stream()), "pin_crop_frame")
    c = int(cnt.item())
    if ts is not None:
        ts_out = ts_out[:c].to(ts.dtype).reshape((c,) + tuple(ts.shape[1:]))
    return out[:c], ts_out


def intrinsic_correct(points: torch.Tensor, correct_deg=0.0):
    if correct_deg == 0.0:
        return points
    if not (points.is_cuda and points.dtype == torch.float32 and points.is_contiguous()):
        raise RuntimeError("intrinsic_correct works in place on a contiguous float32 device tensor")
    check(_lib.lib().pin_intrinsic_correct(points.data_ptr(), points.shape[1], points.shape[0], float(correct_deg),
                                           ops._stream()), "pin_intrinsic_correct")
    return points


def deskewing(points: torch.Tensor, ts: Optional[torch.Tensor], pose: torch.Tensor, ts_mid_pose=0.5):
    if ts is None:
        return points
    # in place on rows of a float32 device tensor; a column slice of wider rows (the reference hands in
    # cur_source_torch[:, :3], slam_dataset.py:478) is fine: the row stride is passed on
    if not (points.is_cuda and points.dtype == torch.float32 and points.dim() == 2 and points.shape[1] >= 3
            and points.stride(1) == 1):
        raise RuntimeError("deskewing works in place on float32 device rows (xyz first, unit column stride)")
    t32 = _dev_f32(ts).reshape(-1).contiguous()
    T = np.ascontiguousarray(pose.detach().to("cpu", torch.float64).numpy() if isinstance(pose, torch.Tensor)
                             else np.asarray(pose, np.float64))
    ws = _ws(64, points.device)
    check(_lib.lib().pin_deskew(points.data_ptr(), points.stride(0), points.shape[0], t32.data_ptr(), T.ctypes.data,
                                float(ts_mid_pose), ws.data_ptr(), ws.numel(), ops._stream()), "pin_deskew")
    return points


def gather(points: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """points[idx] for [n, w] float32 rows through pin_gather_rows."""
    return ops.gather_rows(points, idx if idx.dtype == torch.int32 else idx.to(torch.int32))


class ScanPreprocessor:
    """The data half of SLAMDataset.preprocess_frame for one frame: train-resolution voxel
    down-sampling, crop, optional KITTI correction, source-resolution down-sampling and
    deskewing of the registration source.  Pose bookkeeping stays with the caller."""

    def __init__(self, config):
        self.config = config
        # the whole chain behind ONE C-ABI call with the stage counts on the device and one read-back at the end
        # (pin_preprocess_frame); PIN_PREPROCESS_FUSED=0: stage by stage (three read-backs), the same bits
        import os
        self.fused = os.environ.get("PIN_PREPROCESS_FUSED", "1") != "0"
        self._bufs = None

    def _fused(self, scan, point_ts, train_vox, source_vox, crop_max, last_odom_tran, frame_id, lose_track, enqueue_only=False):
        c, L = self.config, _lib.lib()
        n, w = scan.shape
        if n == 0:
            return None
        dev = scan.device
        ts32 = None if point_ts is None else _dev_f32(point_ts).reshape(-1).contiguous()
        b = self._bufs
        if b is None or b["cap"] < n or b["width"] != w:
            cap = int(n * 1.1) + 1024
            b = self._bufs = dict(cap=cap, width=w,
                                  cnt=torch.zeros((4,), dtype=torch.int32, device=dev), cnt_host=torch.zeros((4,), dtype=torch.int32).pin_memory(),
                                  ws=torch.empty((int(L.pin_preprocess_workspace_bytes(cap, w)) + 256,), dtype=torch.uint8, device=dev))
        pp = _lib.PreprocessParams()
        pp.train_vox, pp.source_vox = float(np.float32(train_vox)), float(np.float32(source_vox))
        pp.min_z, pp.max_z, pp.min_range, pp.max_range = float(c.min_z), float(c.max_z), float(c.min_range), float(crop_max)
        pp.correct_deg = float(c.correction_deg) if getattr(c, "kitti_correction_on", False) else 0.0
        pp.want_source = int(frame_id > 0)
        dsk = bool(getattr(c, "deskew", False) and not lose_track and ts32 is not None and last_odom_tran is not None and frame_id > 0)
        pp.deskew = int(dsk)
        if dsk:
            T = np.ascontiguousarray(last_odom_tran.detach().to("cpu", torch.float64).numpy() if isinstance(last_odom_tran, torch.Tensor)
                                     else np.asarray(last_odom_tran, np.float64)).reshape(-1)
            for i in range(16):
                pp.pose[i] = float(T[i])
        pp.ts_mid_pose = 0.5
        # every frame gets fresh output tensors (the caller keeps pc / source across frames); the scratch is reused
        pc = torch.empty((n, w), dtype=torch.float32, device=dev)
        ts_out = torch.empty((n,), dtype=torch.float32, device=dev) if ts32 is not None else None
        src = torch.empty((n, 3), dtype=torch.float32, device=dev)
        rest = torch.empty((n, w - 3), dtype=torch.float32, device=dev) if (w > 3 and c.color_on) else None
        check(L.pin_preprocess_frame(C.byref(pp), scan.data_ptr(), w, n, None if ts32 is None else ts32.data_ptr(), pc.data_ptr(),
                                     None if ts_out is None else ts_out.data_ptr(), src.data_ptr(), None if rest is None else rest.data_ptr(),
                                     b["cnt"].data_ptr(), b["ws"].data_ptr(), b["ws"].numel(), ops._stream()), "pin_preprocess_frame")
        b["cnt_host"].copy_(b["cnt"], non_blocking=True)
        st = dict(pc=pc, ts_out=ts_out, src=src, rest=rest, point_ts=point_ts, frame_id=frame_id)
        if enqueue_only:  # begin(): the caller waits for the counts later (finish())
            return st
        torch.cuda.current_stream().synchronize()
        return self._fused_finish(st)

    def _fused_finish(self, st):
        """The second half of _fused, once the counts are on the host: the outputs as prefix views."""
        c, b = self.config, self._bufs
        pc, ts_out, src, rest, point_ts, frame_id = (st[k] for k in ("pc", "ts_out", "src", "rest", "point_ts", "frame_id"))
        c1, c2, c3 = (int(v) for v in b["cnt_host"][:3])
        if c1 < 0 or c3 < 0:  # voxel ids too wide for the one-word sort key: the stages one by one (general down-sampling)
            return None
        ts_ret = None
        if point_ts is not None:
            ts_ret = ts_out[:c2].to(point_ts.dtype).reshape((c2,) + tuple(point_ts.shape[1:]))
        # The prefixes are returned as views: pc / src are sized by the raw scan, so a frame's results hold the raw scan's
        # bytes (n x (w + 3 [+ w - 3]) floats: 2.8 MB per 100 000 points) until the caller drops them at the next frame --
        # one frame's worth, never more, against a copy launch per output in a 0.27 ms stage.  PIN_PREPROCESS_COMPACT=1
        # returns compact copies instead (a caller that keeps every frame's cloud).
        compact = os.environ.get("PIN_PREPROCESS_COMPACT", "0") == "1"
        out = (lambda t: t.clone()) if compact else (lambda t: t)
        if frame_id <= 0:
            return out(pc[:c2]), ts_ret, None, None
        # colours: the staged path hands back src[:, 3:] whenever color_on is set -- an empty [c3, 0] tensor for a scan
        # without colour columns, not None
        colors = out(rest[:c3]) if rest is not None else (src.new_empty((c3, 0)) if c.color_on else None)
        return out(pc[:c2]), ts_ret, out(src[:c3]), colors

    def __call__(self, scan: torch.Tensor, point_ts: Optional[torch.Tensor] = None, last_odom_tran=None,
                 frame_id: int = 1, lose_track: bool = False, stream: Optional[torch.cuda.Stream] = None):
        """``stream``: queue the whole chain (and wait for its counts) on THIS stream instead of the current one -- the
        loader's upload stream in a pipeline that reads frame f+1 while frame f's map update is still running on the main
        stream: nothing in the chain reads the map, so it need not queue behind it, and its one read-back then waits for
        0.25 ms of its own launches instead of for the whole main stream.  The caller vouches that ``scan`` / ``point_ts``
        are complete on the device (uploaded on ``stream``, or by work the host has already waited for); the results are
        complete when the call returns and are registered with the caller's current stream (record_stream), so they can
        be used there without an event and their memory is not recycled under it."""
        if getattr(self, "_ticket_open", False):
            raise RuntimeError("ScanPreprocessor: a ticket of begin() is open (its counts sit in this object's buffers): finish() it first")
        if stream is not None and scan.is_cuda:
            user = torch.cuda.current_stream(scan.device)
            if stream != user:
                if isinstance(last_odom_tran, torch.Tensor) and last_odom_tran.is_cuda:
                    last_odom_tran = last_odom_tran.detach().to("cpu", torch.float64)  # (written on the caller's stream: read there)
                with torch.cuda.stream(stream):
                    out = self._run(scan, point_ts, last_odom_tran, frame_id, lose_track)
                    stream.synchronize()  # (the staged path ends in launches; the fused one has waited already)
                for t in out:
                    if isinstance(t, torch.Tensor) and t.is_cuda:
                        t.record_stream(user)
                return out
        return self._run(scan, point_ts, last_odom_tran, frame_id, lose_track)

    def begin(self, scan: torch.Tensor, point_ts: Optional[torch.Tensor] = None, last_odom_tran=None, frame_id: int = 1,
              lose_track: bool = False, stream: Optional[torch.cuda.Stream] = None):
        """The call in two halves, for a loader that has the next scan early: begin() queues the chain on `stream` and returns at
        once with a ticket (it may run on a loader THREAD: the stream context is per thread); finish(ticket) waits for the chain's
        counts and returns what the call returns.  One ticket at a time per object.  Configurations whose set-up needs a read-back
        of its own (adaptive_range_on, random down-sampling) and calls without a stream run whole in finish()."""
        if getattr(self, "_ticket_open", False):
            raise RuntimeError("ScanPreprocessor.begin: the previous ticket has not been finished")
        c = self.config
        args = (scan, point_ts, last_odom_tran, frame_id, lose_track, stream)
        self._ticket_open = True
        if (stream is None or not scan.is_cuda or not self.fused or getattr(c, "adaptive_range_on", False)
                or getattr(c, "rand_downsample", False) or scan.shape[0] == 0):
            return dict(whole=args)
        if isinstance(last_odom_tran, torch.Tensor) and last_odom_tran.is_cuda:
            last_odom_tran = last_odom_tran.detach().to("cpu", torch.float64)
        with torch.cuda.stream(stream):
            self._switch_stream(scan)
            st = self._fused(_dev_f32(scan).contiguous(), point_ts, c.vox_down_m, c.source_vox_down_m, c.max_range, last_odom_tran, frame_id,
                             lose_track, enqueue_only=True)
            ev = torch.cuda.Event()
            ev.record(stream)
        return dict(st=st, event=ev, stream=stream, whole=args)

    def finish(self, ticket):
        self._ticket_open = False
        if "st" not in ticket:
            return self(*ticket["whole"][:5], stream=ticket["whole"][5])
        ticket["event"].synchronize()
        stream = ticket["stream"]
        user = torch.cuda.current_stream(stream.device)
        with torch.cuda.stream(stream):
            out = self._fused_finish(ticket["st"])
        if out is None:  # (voxel ids too wide for the one-word key: the general path, whole)
            return self(*ticket["whole"][:5], stream=stream)
        for t in out:
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(user)
        return out

    def _switch_stream(self, scan):
        if scan.is_cuda:  # the scratch is shared by all calls: a call on another stream than the last one waits for that one first
            cur = torch.cuda.current_stream(scan.device)
            last = getattr(self, "_last_stream", None)
            if last is not None and last != cur:
                last.synchronize()
            self._last_stream = cur

    def _run(self, scan, point_ts, last_odom_tran, frame_id, lose_track):
        c = self.config
        self._switch_stream(scan)
        crop_max = c.max_range
        scan = _dev_f32(scan)
        if getattr(c, "adaptive_range_on", False):  # slam_dataset.py:399-407: crop range from the scan's own extent
            lo, hi = torch.aminmax(scan[:, :2], dim=0)
            (x0, y0), (x1, y1) = lo.abs().tolist(), hi.abs().tolist()
            crop_max = min(c.max_range, 2.0 * max(min(x1, x0), min(y1, y0)))
        train_vox = (crop_max / c.max_range) * c.vox_down_m
        source_vox = (crop_max / c.max_range) * c.source_vox_down_m
        scan = _dev_f32(scan).contiguous()
        if self.fused and not getattr(c, "rand_downsample", False):
            out = self._fused(scan, point_ts, train_vox, source_vox, crop_max, last_odom_tran, frame_id, lose_track)
            if out is not None:
                return out
        if getattr(c, "rand_downsample", False):
            idx = torch.randint(0, scan.shape[0], (int(scan.shape[0] * c.rand_down_r),), device=scan.device)
        else:
            idx = _voxel_down_sample_i32(scan[:, :3], train_vox)
        pc = gather(scan, idx)
        ts = None if point_ts is None else point_ts[idx]
        pc, ts = crop_frame(pc, ts, c.min_z, c.max_z, c.min_range, crop_max)
        if getattr(c, "kitti_correction_on", False):
            pc = intrinsic_correct(pc, c.correction_deg)
        source = source_colors = None
        if frame_id > 0:
            idx2 = _voxel_down_sample_i32(pc[:, :3], source_vox)
            src = gather(pc, idx2)
            source = src[:, :3].contiguous()
            if c.color_on:
                source_colors = src[:, 3:]
            if getattr(c, "deskew", False) and not lose_track and ts is not None and last_odom_tran is not None:
                source = deskewing(source, ts[idx2], torch.as_tensor(last_odom_tran))
        return pc, ts, source, source_colors


def patch_reference():
    """Drop-in mode: rebind the reference's module-level helpers (``dataset.slam_dataset`` and
    ``utils.tools`` must already be importable, i.e. after ``dropin.install``)."""
    import importlib
    sd = importlib.import_module("dataset.slam_dataset")
    tools = importlib.import_module("utils.tools")
    for mod in (sd, tools):
        for name, fn in (("voxel_down_sample_torch", voxel_down_sample_torch), ("crop_frame", crop_frame),
                         ("intrinsic_correct", intrinsic_correct), ("deskewing", deskewing)):
            if hasattr(mod, name):
                setattr(mod, name, fn)
    return sd
