"""Host copies of small device tensors the hot path has to read on the host (the 4x4 pose the tracker just produced
from host numbers, the sensor position cut out of it): a device -> host copy of 128 bytes is a stream synchronisation
of ~50 us, and Mapper.process_frame / NeuralPoints.reset_local_map are handed tensors whose values this process wrote
itself a moment ago.

`remember(t, a)` records that device tensor `t` currently holds the host array `a`; `lookup(t)` returns a copy of `a`
if `t` is that tensor or a view / detach() of it with the same layout and nothing has written to the storage since
(torch's version counter, shared by all views of a storage).  The cache keeps a reference to the tensor, so its memory
cannot be recycled for another tensor while the entry lives: (data_ptr, version) cannot collide.

Freshness rests on torch's version counter, which only torch ops bump: libpinhip kernels and RCCL write through raw
pointers.  Only remember() tensors that torch wrote from host numbers (the pose: torch.tensor(host_array)), and call
invalidate(t) from any code that lets a C-ABI call write into a tensor that may be cached."""
from __future__ import annotations

import numpy as np
import torch

_entries: list = []  # (tensor, version, array), newest last
_CAP = 4


def remember(t: torch.Tensor, a) -> None:
    if not isinstance(t, torch.Tensor) or t.numel() > 64:
        return
    _entries.append((t, t._version, np.array(a, copy=True)))
    del _entries[:-_CAP]


def invalidate(t: torch.Tensor) -> None:
    """Forget every entry that shares storage with `t` (for writers the version counter does not see)."""
    if not isinstance(t, torch.Tensor):
        return
    try:
        base = t.untyped_storage().data_ptr()
    except Exception:
        base = t.data_ptr()
    _entries[:] = [e for e in _entries if e[0].untyped_storage().data_ptr() != base]


def lookup(t: torch.Tensor):
    for u, ver, a in reversed(_entries):
        if (u.data_ptr() == t.data_ptr() and u._version == ver and t._version == ver and u.dtype == t.dtype
                and tuple(u.shape) == tuple(t.shape) and u.stride() == t.stride() and u.device == t.device):
            return a.copy()
    return None


def to_host(t: torch.Tensor) -> np.ndarray:
    """t as a float64-preserving numpy array (its own dtype), through the cache when possible; remembered afterwards."""
    a = lookup(t)
    if a is None:
        a = t.detach().to("cpu").numpy().copy()
        remember(t, a)
    return a


# ---- optional host-side stamps (PIN_HOST_TRACE=1): where the Python time of a frame goes ---------------------------------------
import os as _os
import time as _time

TRACE = _os.environ.get("PIN_HOST_TRACE", "0") == "1"
_stamps: list = []


def stamp(label: str) -> None:
    if TRACE:
        _stamps.append((label, _time.perf_counter()))


def trace_summary():
    """Host time (ms) between consecutive stamps, keyed "from -> to", over everything recorded so far: (mean, count, median)."""
    acc: dict = {}
    for (a, ta), (b, tb) in zip(_stamps, _stamps[1:]):
        acc.setdefault(f"{a} -> {b}", []).append(tb - ta)
    # (mean, count, median: the mean carries the first frames' allocations and the legs with other workloads)
    return {k: (round(1e3 * sum(v) / len(v), 4), len(v), round(1e3 * sorted(v)[len(v) // 2], 4)) for k, v in acc.items()}


def trace_reset():
    del _stamps[:]
