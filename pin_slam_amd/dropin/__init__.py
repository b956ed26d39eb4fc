"""Drop-in mirrors of the reference's hot-path classes.

``install(reference_root)`` makes ``model.neural_points``, ``model.decoder``, ``utils.mapper``,
``utils.tracker`` and ``utils.mesher`` resolve to this package while every other ``model.*`` / ``utils.*`` /
``dataset.*`` module keeps resolving to the reference tree, so the reference's ``pin_slam.py``
runs unchanged.  Nothing is copied from the reference."""
from __future__ import annotations

import importlib
import os
import sys
import types

_OURS = ("model.decoder", "model.neural_points", "utils.tracker", "utils.mapper", "utils.mesher")


def cpu_quota() -> float:
    """CPUs this process may use: the cgroup's quota (cpu.max, v2; cfs_quota_us / cfs_period_us, v1) capped by the
    scheduler affinity.  A GPU box reports every core of the host (256 on the MI355X hosts of this project) while its
    container may be limited to a few of them."""
    n = float(len(os.sched_getaffinity(0))) if hasattr(os, "sched_getaffinity") else float(os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, float(q) / float(p))
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, q / p)
        except (OSError, ValueError):
            pass
    return max(1.0, n)


def limit_host_threads(verbose: bool = True) -> int:
    """torch sizes its intra-op thread pool by the host's core count; the reference's CPU-side code around the hot path (scan-
    context loop detection, logging) then bursts 128 threads into a container quota of 16 CPUs, the cgroup is throttled for
    the rest of the 100 ms period and the NEXT stage of the SLAM loop -- whichever it is -- stalls for ~85 ms although the
    GPU work of a frame is a few ms (measured: profiles/r04_e2e_host_threads.json: 14-25 frames of 60 with a stage above
    3x its median at 128 threads, none at 4).  In drop-in mode the heavy arithmetic is on the GPU: when -- and only when -- the
    container's quota is below the CPUs the scheduler shows, cap the pool at half the quota (the other half is for the
    reference's own helper threads; 16 threads in a quota of 16 still throttled in the measurement).  An unconstrained host keeps
    torch's sizing.  PIN_KEEP_THREADS=1 leaves it alone in every case."""
    import torch
    have = torch.get_num_threads()
    if os.environ.get("PIN_KEEP_THREADS", "0") == "1":
        return have
    affinity = float(len(os.sched_getaffinity(0))) if hasattr(os, "sched_getaffinity") else float(os.cpu_count() or 1)
    quota = cpu_quota()
    if quota >= affinity:  # no cgroup limit below what the scheduler offers: torch's own sizing is right, leave it alone
        return have
    want = max(1, min(have, int(quota // 2) or 1))
    if want < have:
        torch.set_num_threads(want)
        if verbose:
            print(f"[pin_slam_amd] torch intra-op threads {have} -> {want} (CPU quota of this container: {cpu_quota():.0f}; "
                  f"PIN_KEEP_THREADS=1 to keep {have})", flush=True)
    return want


def install(reference_root: str):
    """Wire the module namespace for drop-in use.  Call before importing pin_slam."""
    reference_root = os.path.abspath(reference_root)
    if not os.path.isdir(os.path.join(reference_root, "utils")):
        raise FileNotFoundError(f"no PIN-SLAM tree at {reference_root}")
    if reference_root not in sys.path:
        sys.path.insert(0, reference_root)
    here = os.path.dirname(os.path.abspath(__file__))
    for pkg in ("model", "utils"):
        for k in [k for k in sys.modules if k == pkg or k.startswith(pkg + ".")]:
            del sys.modules[k]
        m = types.ModuleType(pkg)
        # our directory first, the reference's second: same-named modules resolve to ours,
        # everything else (utils.config, utils.tools, utils.mesher, ...) to the reference
        m.__path__ = [os.path.join(here, pkg), os.path.join(reference_root, pkg)]
        m.__package__ = pkg
        sys.modules[pkg] = m
    limit_host_threads()
    out = {}
    for name in _OURS:  # order matters: utils.mapper looks for the reference Mapper to inherit
        real = importlib.import_module("pin_slam_amd.dropin." + name)
        sys.modules[name] = real
        setattr(sys.modules[name.split(".")[0]], name.split(".")[1], real)
        out[name] = real
    return out
