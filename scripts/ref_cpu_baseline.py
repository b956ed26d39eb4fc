#!/usr/bin/env python3
"""The REAL reference (PRBonn/PIN_SLAM, torch CPU) timed on the bench's C3 inputs: BASELINE.md section 3 protocol.

Runs where the reference tree is: /root/reference in the build container, or -- ON THE GPU BOX'S HOST, which is where
BASELINE.md section 3 wants the number from -- the git-ignored pack `python scripts/e2e_pin_slam.py pack` leaves under
oracle/_ref/ (it travels with a gpurun snapshot; unpacked into a temporary directory here, PIN_REFERENCE_ROOT).  Loads the
unmodified NeuralPoints / Decoder / Tracker / Mapper through oracle/ref_loader.py, builds the synthetic C3 map of bench.py
(pin_slam_amd.synth: ~2.2 M neural points, 5e7-slot table, Kc = 81, k = 8, decoder 4x64) inside the reference's
own classes and times

  * Tracker.registration_step  on the 100k-point scan (one Gauss-Newton step: query_source_points + implicit_reg),
  * Tracker.tracking           (reg_iter_n = 50, the loop with its own convergence test),
  * Mapper.mapping(12)         (batch 16384 + Eikonal, backward, Adam over every local feature),

with time.perf_counter, 1 warm-up + the median of `--reps` repeats, torch.get_num_threads() threads.  Writes
profiles/r04_ref_cpu_baseline.json (--out), which bench.py emits as `cpu_baseline` (kind "reference-torch-cpu";
baseline only: a GPU/CPU ratio says nothing about kernel quality).  No GPU is touched."""
from __future__ import annotations

import argparse
import json
import os
import platform
import statistics
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PACK = os.path.join(ROOT, "oracle", "_ref", "pin_slam_reference.tar.gz")
if not os.path.isdir(os.environ.get("PIN_REFERENCE_ROOT", "/root/reference")) and os.path.exists(PACK):
    import tarfile
    import tempfile
    _dst = tempfile.mkdtemp(prefix="pin_slam_ref_")
    with tarfile.open(PACK) as _tf:
        _tf.extractall(_dst)
    os.environ["PIN_REFERENCE_ROOT"] = _dst
from oracle import ref_loader as R  # noqa: E402
from pin_slam_amd import synth  # noqa: E402


def host_string(label):
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    gpu = os.path.exists("/dev/kfd") and torch.cuda.is_available()
    try:
        aff = len(os.sched_getaffinity(0))
    except AttributeError:
        aff = os.cpu_count()
    what = label or ("gpu box (MI355X host)" if gpu else "build container (no GPU)")
    return f"{what}: {platform.node()}, {os.cpu_count()} logical CPUs ({aff} usable), {model or platform.machine()}"


def timed(fn, reps, warmup=True):
    if warmup:
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts), ts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--layers", type=int, default=16)
    ap.add_argument("--scan", type=int, default=100_000)
    ap.add_argument("--bs", type=int, default=16384)
    ap.add_argument("--map-iters", type=int, default=12)
    ap.add_argument("--reg-iters", type=int, default=50)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r04_ref_cpu_baseline.json"))
    ap.add_argument("--host-label", default="", help="what to call this machine in the record (e.g. 'gpu box (MI355X host)')")
    ap.add_argument("--skip-tracking", action="store_true", help="do not time Tracker.tracking (a thread-count sweep needs the two "
                                                                 "quantities of the frame only)")
    ap.add_argument("--tracking-reps", type=int, default=0, help="repeats of Tracker.tracking (0 = --reps); 1 = ONE call without a "
                                                                 "warm-up (the bench's live leg: a call is ~50 registration steps)")
    ap.add_argument("--threads", type=int, default=0, help="torch.set_num_threads (0 = torch's default for this host)")
    a = ap.parse_args()
    if a.threads > 0:
        torch.set_num_threads(a.threads)
    m = R.load()
    H, L, k = 64, 4, 8
    cfg = R.make_config(voxel_size_m=0.4, search_alpha=0.5, num_nei_cells=2, query_nn_k=k, buffer_size=int(5e7),
                        feature_std=0.1, bs=a.bs, max_range=80.0, local_map_radius=82.0, local_map_travel_dist_ratio=5.0,
                        track_on=True, reg_iter_n=a.reg_iters, weighted_first=True)
    cfg.geo_mlp_level, cfg.geo_mlp_hidden_dim = L, H
    torch.manual_seed(42)
    sm = synth.build_map(layers=a.layers)
    dec = m["Decoder"](cfg, H, L, 1)
    npts = m["NeuralPoints"](cfg)
    npts.travel_dist = torch.zeros(8, dtype=torch.float32)
    t0 = time.perf_counter()
    npts.update(torch.from_numpy(sm.positions), torch.zeros(3), torch.eye(3), 0)
    t_update = time.perf_counter() - t0
    P = npts.count()
    ds = R.FakeDataset(n_frames=4)
    mp = m["Mapper"](cfg, ds, npts, {"sdf": dec, "semantic": None, "color": None})
    trk = m["Tracker"](cfg, npts, {"sdf": dec, "semantic": None, "color": None})
    pool_c, pool_l = synth.make_pool(sm, n=2_000_000)
    mp.coord_pool = mp.global_coord_pool = torch.from_numpy(pool_c)
    mp.sdf_label_pool = torch.from_numpy(pool_l)
    mp.weight_pool = torch.ones(len(pool_l))
    mp.time_pool = torch.zeros(len(pool_l), dtype=torch.int)
    mp.pool_sample_count = len(pool_l)
    mp.sem_label_pool = mp.color_pool = mp.normal_label_pool = None  # (init_pool leaves empty tensors)
    mp.determine_used_pose()
    # a few training iterations so that the field has gradients of sensible size (not timed)
    mp.mapping(20)
    scan = torch.from_numpy(synth.make_scan(sm, n=a.scan, seed=1))
    ang = 0.003
    T_init = torch.eye(4, dtype=torch.float64)
    T_init[:3, :3] = torch.tensor([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]])
    T_init[:3, 3] = torch.tensor([0.05, -0.04, 0.02])
    src = m["tools"].transform_torch(scan, T_init)

    zeros = torch.zeros(src.shape[0])  # source_sdf, as Tracker.tracking passes it (tracker.py:96-97)

    def reg_step():
        trk.registration_step(src, None, zeros, None, cfg.reg_min_grad_norm, cfg.reg_max_grad_norm, cfg.reg_GM_dist_m,
                              cfg.reg_GM_grad, cfg.reg_lm_lambda)

    def tracking():
        trk.tracking(scan, T_init)

    def mapping():
        mp.mapping(a.map_iters)

    out = {"what": "unmodified PRBonn/PIN_SLAM classes on torch CPU (oracle/ref_loader.py), bench.py C3 inputs",
           "host": host_string(a.host_label),
           "torch_threads": torch.get_num_threads(), "torch": torch.__version__, "neural_points": int(P),
           "scan_points": a.scan, "decoder": f"{L}x{H}", "knn_k": k, "candidate_cells": int(npts.neighbor_K),
           "map_build_s": round(t_update, 2), "reps": a.reps}
    t, all_ = timed(reg_step, a.reps)
    out["registration_step_ms"] = round(t * 1e3, 1)
    out["registration_step_ms_all"] = [round(x * 1e3, 1) for x in all_]
    out["registration_queries_per_sec"] = round(a.scan / t, 1)
    print("registration_step", out["registration_step_ms"], "ms", [round(x * 1e3) for x in all_], flush=True)
    t, all_ = timed(mapping, a.reps)
    out["mapping_ms"] = round(t * 1e3, 1)
    out["mapping_ms_all"] = [round(x * 1e3, 1) for x in all_]
    out["mapping_iterations"] = a.map_iters
    out["mapper_samples_per_sec"] = round(a.bs * a.map_iters / t, 1)
    print("mapping", out["mapping_ms"], "ms", [round(x * 1e3) for x in all_], flush=True)
    if not a.skip_tracking:
        tr = a.tracking_reps or a.reps
        t, all_ = timed(tracking, tr, warmup=tr > 1)
        out["tracking_ms"] = round(t * 1e3, 1)
        out["tracking_ms_all"] = [round(x * 1e3, 1) for x in all_]
        out["tracking_reps"] = tr
        out["tracking_note"] = "Tracker.tracking with its own convergence test (it may stop before reg_iter_n iterations)"
        print("tracking", out["tracking_ms"], "ms", [round(x * 1e3) for x in all_], flush=True)
    # the bench's frame: reg_iters GN steps without early exit over the whole scan + map_iters mapping iterations
    frame_s = a.reg_iters * out["registration_step_ms"] / 1e3 + out["mapping_ms"] / 1e3
    out["frames_per_sec_bench_definition"] = round(1.0 / frame_s, 5)
    # the spread of the frame figure: every repeat of the two quantities, fastest with fastest and slowest with slowest
    lo = a.reg_iters * min(out["registration_step_ms_all"]) / 1e3 + min(out["mapping_ms_all"]) / 1e3
    hi = a.reg_iters * max(out["registration_step_ms_all"]) / 1e3 + max(out["mapping_ms_all"]) / 1e3
    out["frames_per_sec_range"] = [round(1.0 / hi, 5), round(1.0 / lo, 5)]
    out["frame_definition"] = (f"{a.reg_iters} x registration_step over the whole {a.scan}-point scan (no early exit) + "
                               f"Mapper.mapping({a.map_iters}); preprocess / map-prep stages not included")
    out["kind"] = "reference-torch-cpu"
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
