#!/usr/bin/env python3
"""bench.py -- PIN-SLAM hot path on MI355X: SLAM frames/s (+ mapper samples/s).

One "step" = one SLAM frame on the synthetic workload of SURVEY.md 8(d), config C3:
  * odometry: Tracker.tracking = `reg_iters` Gauss-Newton iterations (reference default
    reg_iter_n = 50, no early exit => worst case) over a 100k-point scan against a
    ~2.2M-neural-point map (kNN=8, Kc=81, decoder 4x64), pose read back every iteration;
  * mapping: Mapper.mapping = 12 iterations, batch 16384 (+ 6*1639 Eikonal queries), BCE +
    Eikonal, backward to features and decoder, dense Adam over all local features.
Inputs are resident in HBM before the timed region.  Preprocessing / map growth
(Mapper.process_frame, SURVEY 8f "next" rows) are not part of the step yet.

N > 1 (torchrun, one rank per GPU): registration is "replicas only" (each rank registers its
own scan); the mapper batch is N x 16384, sharded, with one RCCL all-reduce of
[decoder grads | feature grads] per iteration (weak scaling).  value = N * frames / time.

Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: layers, hidden, levels
    "c3": dict(layers=16, hidden=64, levels=4, desc="100k-pt scan, ~2.2M neural points, kNN=8, Kc=81, decoder 4x64"),
    "c2": dict(layers=4, hidden=32, levels=2, desc="100k-pt scan, ~0.56M neural points, kNN=8, Kc=81, decoder 2x32"),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c3", choices=list(WORKLOADS))
    ap.add_argument("--reg-iters", type=int, default=50)
    ap.add_argument("--map-iters", type=int, default=12)
    ap.add_argument("--bs", type=int, default=16384)
    ap.add_argument("--scan", type=int, default=100_000)
    ap.add_argument("--pretrain-iters", type=int, default=300)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sort-scan", type=int, default=1, help="voxel-order the scan once per frame")
    ap.add_argument("--events", default="all", choices=["none", "knn", "all"],
                    help="HIP events around the tracker's kNN / GN launches inside the timed region")
    ap.add_argument("--bricks", type=int, default=1, help="per-frame brick cache for the kNN (identical results)")
    return ap.parse_args()


def dev(a, dt=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dt is not None:
        t = t.to(dt)
    return t.cuda()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    from pin_slam_amd import engine, ops, synth
    from pin_slam_amd._lib import GnParams

    wl = WORKLOADS[args.workload]
    H, L, k = wl["hidden"], wl["levels"], 8
    res, sigma_sigmoid = 0.4, 0.1
    sdf_scale = 0.55 * sigma_sigmoid

    # ---------------- synthetic map / scan / pool (identical on every rank) ----------------
    m = synth.build_map(layers=wl["layers"], resolution=res)
    P = len(m.positions)
    pos = dev(m.positions)
    ts_create = torch.zeros(P, dtype=torch.int32, device="cuda")
    pos4 = torch.empty((P, 4), dtype=torch.float32, device="cuda")
    ops.pack_positions(pos, ts_create, pos4)
    dx, mv = ops.search_neighborhood(2, 0.5, res)
    g2l = torch.arange(P + 1, dtype=torch.int32, device="cuda")
    g2l[-1] = -1
    st = ops.SearchState(table=dev(m.table), pos4=pos4, cand_off=dev(ops.candidate_offsets(dx, m.buffer_size)),
                         n_points=P, resolution=res, max_valid_dist2=mv,
                         travel_dist=torch.zeros(1, dtype=torch.float32, device="cuda"), cur_ts=0,
                         diff_travel_dist_local=82.0 * 5.0, global2local=g2l)
    feats = dev(m.features)
    dec = dev(synth.init_decoder(H, L))
    cert = torch.zeros(P, dtype=torch.float32, device="cuda")
    ts_update = torch.zeros(P, dtype=torch.int32, device="cuda")
    fs = ops.FieldState(feats=feats, dec=dec, k=k, hidden=H, levels=L, weighted_first=True, sdf_scale=sdf_scale,
                        certainty=cert, pos=pos)
    scan_np = synth.make_scan(m, n=args.scan, seed=1 + rank)
    scan = dev(scan_np)
    if args.sort_scan:  # once per frame in a real run; part of preprocessing, not of the GN loop
        key = torch.floor(scan / res).long()
        k2 = (key[:, 0] + 4096) + ((key[:, 1] + 4096) << 14) + ((key[:, 2] + 4096) << 28)
        scan = scan[torch.argsort(k2)].contiguous()
    pool_c, pool_l = synth.make_pool(m, n=2_000_000)
    pool = (dev(pool_c), dev(pool_l), torch.ones(len(pool_l), dtype=torch.float32, device="cuda"),
            torch.zeros(len(pool_l), dtype=torch.int32, device="cuda"))
    bs_global = args.bs * world
    trainer = engine.MapTrainer(st, fs, *pool, ts_update, bs=bs_global, decimation=10, sigma=sdf_scale, weight_e=0.5,
                                eik_eps=res * 0.2, rank=rank, world=world)
    gen = torch.Generator().manual_seed(1)

    def batches(n_iter):
        idx = torch.randint(0, len(pool_l), (n_iter, bs_global), generator=gen, dtype=torch.int64).to(torch.int32)
        sh = idx[:, rank * args.bs:(rank + 1) * args.bs].contiguous().cuda()
        return [sh[i] for i in range(n_iter)]

    # pre-train so that the SDF is a real field (GN accepts points); not timed
    for _ in range(max(1, args.pretrain_iters // 50)):
        trainer.mapping(batches(50))
    gp = GnParams()
    gp.valid_nn_k, gp.min_grad_norm, gp.max_grad_norm = k, 0.5, 2.0
    gp.max_sdf_std, gp.gm_dist, gp.gm_grad = 0.25, 0.3, 0.1
    tracker = engine.GNTracker(st, fs, gp, 1e-4, args.scan)
    ang = 0.003
    T_init = np.eye(4)
    T_init[:3, :3] = [[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]]
    T_init[:3, 3] = [0.05, -0.04, 0.02]

    # HIP events around every GN-accumulate launch (the dominant kernel of the frame) and every kNN
    # launch of the tracker, recorded on the launch stream inside the timed region
    ev_pairs, gn_pairs = [], []
    cur_ev = {}

    ev_frames = min(2, args.steps)  # instrument the first timed frames only (events cost ~3 % each)
    n_ev = ev_frames * args.reg_iters
    pool = {t: [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_ev)]
            for t in ("k", "g")}  # created (and warmed) outside the timed region
    for t in pool:
        for a_, b_ in pool[t]:
            a_.record(); b_.record()
    torch.cuda.synchronize()

    def bracket(store, tag):
        def hook(start):
            i = len(store) if start else len(store)
            if start:
                cur_ev[tag] = pool[tag][len(store)]
                cur_ev[tag][0].record()
            else:
                cur_ev[tag][1].record()
                store.append(cur_ev[tag])
        return hook

    on_knn = bracket(ev_pairs, "k") if args.events in ("knn", "all") else None
    on_gn = bracket(gn_pairs, "g") if args.events == "all" else None

    bricks = ops.BrickCache(dx, 2) if args.bricks else None
    frame_batches = [batches(args.map_iters) for _ in range(args.warmup + args.steps)]
    stats = {}

    def frame(i, timed):
        t0 = time.perf_counter()
        if bricks is not None:  # rebuilt every frame, as reset_local_map does after each map update
            bricks.build(st)
            tracker.bricks = trainer.bricks = bricks
        T, cnt, res_cm, its, _, _ = tracker.track(scan, T_init, args.reg_iters, early_exit=False)
        t1 = time.perf_counter()
        trainer.mapping(frame_batches[i])
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if timed:
            stats.setdefault("track", []).append(t1 - t0)
            stats.setdefault("map", []).append(t2 - t1)
        stats["last"] = (T, cnt, res_cm, its)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        frame(i, False)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        tracker.on_knn, tracker.on_gn = (on_knn, on_gn) if i < ev_frames else (None, None)
        frame(args.warmup + i, True)
    barrier()
    elapsed = time.perf_counter() - t0
    tracker.on_knn = tracker.on_gn = None
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    knn_ms = float(np.mean([a.elapsed_time(b) for a, b in ev_pairs])) if ev_pairs else float("nan")
    gn_ms = float(np.mean([a.elapsed_time(b) for a, b in gn_pairs])) if gn_pairs else float("nan")
    # fused SDF + Jacobian + GN kernel: decoder flops per query, forward + input Jacobian
    flops_q = 2 * 2 * (11 * H + (L - 1) * H * H + H)
    gn_tflops = flops_q * args.scan / (gn_ms * 1e-3) / 1e12
    FP32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 vector = fp32 MFMA rate
    nn_mean = float(tracker.nn[:args.scan].float().mean().item())
    Kc = int(st.cand_off.numel())
    rho = nn_mean / Kc
    # algorithmic bytes of one kNN launch (DESIGN.md "kernel: knn_query"): query in + out,
    # one 4-byte slot per candidate cell, one 16-byte position per occupied cell, kNN record out
    bytes_q = 12 + 12 + 4 * Kc + 16 * rho * Kc + 16 * k + 4
    achieved = bytes_q * args.scan / (knn_ms * 1e-3) / 1e9
    pmc_data = {}
    pmc = os.path.join(ROOT, "profiles", "r01_pmc.json")
    if os.path.exists(pmc):
        try:
            pmc_data = json.load(open(pmc))
        except Exception:
            pmc_data = {}

    def pmc_get(key):
        return pmc_data.get(key)

    traffic = pmc_get("knn_brick_hbm_bytes_per_launch" if args.bricks else "knn_hbm_bytes_per_launch")

    frames_per_s = world * args.steps / elapsed
    q_total = trainer.buf.Q
    out = {
        "metric": "slam_frames_per_sec", "value": round(frames_per_s, 3), "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {wl['desc']}; frame = {args.reg_iters} GN iterations (no early exit) "
                               f"+ {args.map_iters} mapping iterations of batch {args.bs} (+{trainer.buf.n_eik}x6 Eikonal)",
                   "neural_points": P, "scan_points": args.scan, "knn_k": k, "candidate_cells": Kc,
                   "decoder": f"{L}x{H}", "occupancy_rho": round(rho, 4), "scan_voxel_sorted": bool(args.sort_scan), "brick_cache": bool(args.bricks),
                   "parallelism": "1 GPU" if world == 1 else f"tracker replicas x{world}, mapper dp{world} (RCCL all-reduce)"},
        "mapper_samples_per_sec": round(world * args.bs * args.map_iters / float(np.mean(stats["map"])), 1),
        "tracker_ms_per_frame": round(1e3 * float(np.mean(stats["track"])), 3),
        "mapper_ms_per_frame": round(1e3 * float(np.mean(stats["map"])), 3),
        "gn_valid_points": int(stats["last"][1]), "gn_residual_cm": round(float(stats["last"][2]), 4),
        "roofline": {"kernel": "gn_accumulate_mfma_kernel", "bound": "mfma", "achieved": round(gn_tflops, 2),
                     "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(gn_tflops / FP32_PEAK_TFLOPS, 4),
                     "traffic": pmc_get("gn_hbm_bytes_per_launch"), "avg_launch_ms": round(gn_ms, 4),
                     "launches": len(gn_pairs), "algorithmic_flops_per_query": flops_q,
                     "share_of_frame": round(gn_ms * args.reg_iters / (1e3 * elapsed / args.steps), 3)},
        "roofline_knn": {"kernel": "knn_brick_kernel" if args.bricks else "knn_query_kernel", "bound": "hbm",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "avg_launch_ms": round(knn_ms, 4), "launches": len(ev_pairs),
                         "algorithmic_bytes_per_query": round(bytes_q, 1),
                         "share_of_frame": round(knn_ms * args.reg_iters / (1e3 * elapsed / args.steps), 3)},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(m, scan_np, pool_c, pool_l, feats.cpu().numpy(), dec.cpu().numpy(),
                                           H, L, k, sdf_scale, dx, mv, args)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def cpu_baseline(m, scan, pool_c, pool_l, feats, dec, H, L, k, sdf_scale, dx, mv, args):
    """The numpy oracle (a port of the reference's torch-CPU chain) timed on a bounded sample of
    the same workload on this host; the frame rate is extrapolated linearly in the query count."""
    from oracle import pin_oracle as O
    n_s, bs_s = 20000, 4096
    params = O.unpack_decoder(dec, 11, H, L)
    dx64 = dx.astype(np.int64)
    q = scan[:n_s]

    def reg_step():
        s = O.radius_search(q, m.table, m.positions, m.resolution, dx64, mv)
        sdf, grad, std, nn, _ = O.query_sdf(q, s, feats, m.positions, params, sdf_scale, k, dtype=np.float32)
        O.registration_step(q, sdf, grad, std, nn, valid_nn_k=k)

    def train_iter():
        def searcher(p):
            s = O.radius_search(p, m.table, m.positions, m.resolution, dx64, mv)
            return O.query_feature(p, s, feats, m.positions, None, k, weighted_first=False)
        O.train_step(pool_c[:bs_s], pool_l[:bs_s], np.ones(bs_s, np.float32), searcher, feats, m.positions, dec,
                     (11, H, L), sdf_scale, k, dec=10, eps=0.08, dtype=np.float32)

    from threadpoolctl import threadpool_limits
    limiter = threadpool_limits(limits=1)  # scalar port: one host thread, stated in `cores`
    reg_step()
    t0 = time.perf_counter(); reps = 0
    while time.perf_counter() - t0 < 8.0:
        reg_step(); reps += 1
    t_reg = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter(); reps2 = 0
    while time.perf_counter() - t0 < 8.0:
        train_iter(); reps2 += 1
    t_tr = (time.perf_counter() - t0) / reps2
    limiter.restore_original_limits()
    frame_s = args.reg_iters * t_reg * (args.scan / n_s) + args.map_iters * t_tr * (args.bs / bs_s)
    return {"value": round(1.0 / frame_s, 5), "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"numpy oracle: {reps} registration steps on {n_s} scan points ({t_reg*1e3:.0f} ms each) and "
                      f"{reps2} mapping iterations of batch {bs_s} ({t_tr*1e3:.0f} ms each), extrapolated linearly to "
                      f"{args.reg_iters}x{args.scan} + {args.map_iters}x{args.bs}; one host thread "
                      f"(host has {os.cpu_count()} cores)",
            "registration_queries_per_sec": round(n_s / t_reg, 1), "mapper_samples_per_sec": round(bs_s / t_tr, 1)}


if __name__ == "__main__":
    main()
