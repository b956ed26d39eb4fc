"""CPU tests of bench.py's host-side helpers (no GPU, no timing)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def test_host_description_names_cpu_and_count():
    import bench
    h = bench._host_now()
    assert set(h) == {"model", "cpus", "text"} and h["cpus"] >= 1 and str(h["cpus"]) in h["text"]


def test_live_reference_baseline_is_optional(monkeypatch, tmp_path):
    """cpu_baseline is timed by the bench command only when the reference pack travels with the snapshot (oracle/_ref/) and
    PIN_BENCH_REF_LIVE is not 0; otherwise the helper returns None and the committed record stays the baseline -- it must
    never raise (a bench line without the live leg is still a bench line)."""
    import bench
    args = argparse.Namespace(reg_iters=50, map_iters=12)
    monkeypatch.setenv("PIN_BENCH_REF_LIVE", "0")
    assert bench._ref_cpu_live(args) is None
    monkeypatch.delenv("PIN_BENCH_REF_LIVE")
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))  # a tree without oracle/_ref/
    assert bench._ref_cpu_live(args) is None


def test_thread_cap_only_under_a_container_quota(monkeypatch):
    """dropin.limit_host_threads: torch's intra-op pool is capped only when the cgroup's CPU quota is below the CPUs
    the scheduler shows; an unconstrained host keeps torch's own sizing."""
    import torch
    from pin_slam_amd import dropin
    have = torch.get_num_threads()
    try:
        aff = float(len(os.sched_getaffinity(0))) if hasattr(os, "sched_getaffinity") else float(os.cpu_count() or 1)
        monkeypatch.setattr(dropin, "cpu_quota", lambda: aff)  # no limit below the affinity: leave the pool alone
        assert dropin.limit_host_threads(verbose=False) == have and torch.get_num_threads() == have
        monkeypatch.setattr(dropin, "cpu_quota", lambda: 2.0 if aff > 2 else aff)
        want = dropin.limit_host_threads(verbose=False)
        if aff > 2:
            assert want == 1 and torch.get_num_threads() == 1  # half the quota of 2
        monkeypatch.setenv("PIN_KEEP_THREADS", "1")
        torch.set_num_threads(have)
        assert dropin.limit_host_threads(verbose=False) == have
    finally:
        torch.set_num_threads(have)
