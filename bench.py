#!/usr/bin/env python3
"""bench.py -- PIN-SLAM per-frame pipeline on MI355X: SLAM frames/s (+ mapper samples/s).

One "step" = one SLAM frame, the reference's own frame definition (pin_slam.py:500-502:
preprocess + odometry + mapping preparation + mapping), on the synthetic workload of SURVEY.md
8(d), config C3 (100k-point scan, ~2.2M neural points, kNN=8, Kc=81, decoder 4x64):
  * preprocess   : SLAMDataset.preprocess_frame data path -- voxel down-sampling (vox_down_m),
                   crop_frame, source down-sampling (source_vox_down_m), deskewing;
  * odometry     : Tracker.tracking = `reg_iters` Gauss-Newton iterations (reference default 50,
                   NO early exit => worst case) registering the WHOLE cropped scan (~100k points;
                   the reference registers only the source-down-sampled subset -- that cheaper
                   variant is reported beside it as frames_per_sec_source_downsampled);
  * map prep     : Mapper.process_frame -- 7 samples per ray into the pool, NeuralPoints.update +
                   reset_local_map (+ brick cache), pool window / capacity filter over ~2.7M
                   samples, query_certainty, new-sample index;
  * mapping      : Mapper.mapping = 12 iterations, batch 16384 (+ 6*1639 Eikonal queries), BCE +
                   Eikonal, backward to features and decoder, dense Adam over all local features.
Everything runs through the drop-in classes (pin_slam_amd.dropin) on libpinhip.  Inputs (raw
scan, timestamps) are resident in HBM before the timed region.

N > 1 (torchrun, one rank per GPU): preprocess / odometry / map prep are replicas (every rank
keeps the identical map: same scan, same seed); the mapper batch is N x 16384, sharded, with one
RCCL all-reduce of [decoder grads | feature grads] per iteration (weak scaling).
value = N * frames / time.

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
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 vector = fp32 MFMA rate
BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA


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
    ap.add_argument("--pool", type=int, default=2_000_000, help="samples in the pool (= pool_capacity)")
    ap.add_argument("--pretrain-iters", type=int, default=300)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--parallel", default="auto", choices=["auto", "replicas", "dp"],
                    help="N > 1: 'replicas' = N independent frame streams, no data-path collective; 'dp' = the mapper "
                         "shards a global batch of N x bs and all-reduces [decoder | feature] gradients over RCCL every "
                         "iteration (SURVEY 8e); auto = dp from a per-rank batch of 2^17 up, where the sharded compute "
                         "outweighs the 71 MB gradient exchange, replicas below")
    ap.add_argument("--events", default="all", choices=["none", "knn", "all"],
                    help="HIP events around the tracker's kNN / GN launches inside the timed region")
    ap.add_argument("--stages", default="all", choices=["all", "hot"],
                    help="hot = odometry + mapping only (the r01 a-i bench lines)")
    return ap.parse_args()


class Dataset:
    """The attributes Mapper touches on its dataset (mapper.py:141-159, 212, 457-458)."""
    lose_track = False
    stop_status = False
    static_mask = None
    gt_pose_provided = True

    def __init__(self, n):
        self.processed_frame = 0
        self.odom_poses = np.tile(np.eye(4), (n, 1, 1))
        self.pgo_poses = self.gt_poses = self.odom_poses


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

    from pin_slam_amd import preprocess, synth
    from pin_slam_amd.config import PinConfig
    from pin_slam_amd.dropin.model.decoder import Decoder
    from pin_slam_amd.dropin.model.neural_points import NeuralPoints
    from pin_slam_amd.dropin.utils.mapper import Mapper
    from pin_slam_amd.dropin.utils.tracker import Tracker

    wl = WORKLOADS[args.workload]
    H, L, k = wl["hidden"], wl["levels"], 8
    mapper_dp = world > 1 and (args.parallel == "dp" or (args.parallel == "auto" and args.bs >= (1 << 17)))
    dp_world = world if mapper_dp else 1
    res = 0.4
    n_frames = args.warmup + 2 * args.steps + 4
    cfg = PinConfig(voxel_size_m=res, search_alpha=0.5, num_nei_cells=2, query_nn_k=k, buffer_size=int(5e7),
                    feature_std=0.1, bs=args.bs * dp_world, iters=args.map_iters, max_range=80.0, local_map_radius=82.0,
                    window_radius=80.0, local_map_travel_dist_ratio=5.0, vox_down_m=0.08, source_vox_down_m=0.8,
                    min_range=2.5, min_z=-5.0, max_z=80.0, deskew=True, pool_capacity=args.pool, pool_filter_freq=1,
                    bs_new_sample=2048, geo_mlp_level=L, geo_mlp_hidden_dim=H, reg_iter_n=args.reg_iters)
    torch.manual_seed(42)  # identical on every rank: the replicas must keep identical maps and batches

    # ---------------- synthetic map / scan / pool (identical on every rank) ----------------
    m = synth.build_map(layers=wl["layers"], resolution=res)
    npts = NeuralPoints(cfg)
    npts.travel_dist = torch.zeros(n_frames + 1, dtype=torch.float32, device="cuda")
    npts.update(torch.from_numpy(m.positions).cuda(), torch.zeros(3), torch.eye(3), 0)
    P = npts.count()
    dec = Decoder(cfg, H, L, 1)
    decoders = {"sdf": dec, "semantic": None, "color": None}
    ds = Dataset(n_frames + 1)
    mp = Mapper(cfg, ds, npts, decoders)
    mp.dp_rank, mp.dp_world = (rank, world) if mapper_dp else (0, 1)
    trk = Tracker(cfg, npts, decoders)
    pool_c, pool_l = synth.make_pool(m, n=args.pool)
    mp.coord_pool = torch.from_numpy(pool_c).cuda()
    mp.global_coord_pool = mp.coord_pool.clone()  # poses are identity in this workload
    mp.sdf_label_pool = torch.from_numpy(pool_l).cuda()
    mp.weight_pool = torch.ones(len(pool_l), dtype=torch.float32, device="cuda")
    mp.time_pool = torch.zeros(len(pool_l), dtype=torch.int32, device="cuda")
    mp.pool_sample_count = len(pool_l)
    mp._pool()          # adopt the tensors into the device pool
    mp._publish_pool()
    scan_np = synth.make_scan(m, n=args.scan, seed=1)
    rng = np.random.default_rng(5)
    raw = torch.from_numpy(np.concatenate([scan_np, rng.random((args.scan, 1), dtype=np.float32)], 1)).cuda()
    raw_ts = torch.from_numpy(rng.random(args.scan, dtype=np.float32)).cuda()
    last_odom = np.eye(4)
    last_odom[:3, 3] = [0.5, 0.0, 0.0]

    # pre-train so that the SDF is a real field (GN accepts points); not timed
    for _ in range(max(1, args.pretrain_iters // 50)):
        mp.mapping(50)
    ang = 0.003
    T_init = np.eye(4)
    T_init[:3, :3] = [[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]]
    T_init[:3, 3] = [0.05, -0.04, 0.02]
    gp = trk._gn_params(cfg.reg_min_grad_norm, cfg.reg_max_grad_norm, cfg.reg_GM_dist_m, cfg.reg_GM_grad)
    prep = preprocess.ScanPreprocessor(cfg)
    pose_t = torch.eye(4, dtype=torch.float64, device="cuda")

    # HIP events around every GN-accumulate launch (the dominant kernel of the frame) and every kNN
    # launch of the tracker, recorded on the launch stream inside the timed region
    ev_pairs, gn_pairs = [], []
    cur_ev = {}
    ev_frames = min(2, args.steps)  # instrument the first timed frames only (events cost ~3 % each)
    n_ev = ev_frames * args.reg_iters
    evpool = {t: [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_ev)]
              for t in ("k", "g")}  # created (and warmed) outside the timed region
    for t in evpool:
        for a_, b_ in evpool[t]:
            a_.record(); b_.record()
    torch.cuda.synchronize()

    def bracket(store, tag):
        def hook(start):
            if start:
                cur_ev[tag] = evpool[tag][len(store)]
                cur_ev[tag][0].record()
            else:
                cur_ev[tag][1].record()
                store.append(cur_ev[tag])
        return hook

    on_knn = bracket(ev_pairs, "k") if args.events in ("knn", "all") else None
    on_gn = bracket(gn_pairs, "g") if args.events == "all" else None
    stats = {}
    state = {"fid": 1, "cloud": None, "src": None}

    stage_events = []  # per timed frame: 5 events at the stage boundaries (no host sync between the stages)

    def frame(timed, hooks=(None, None), source_downsampled=False):
        fid = state["fid"]
        state["fid"] += 1
        ds.processed_frame = fid
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)] if timed else None
        if ev: ev[0].record()
        if args.stages == "all" or state["cloud"] is None:
            pc, _, src, _ = prep(raw, raw_ts, last_odom_tran=last_odom, frame_id=fid)
            state["cloud"], state["src"] = pc, src
            state["xyz"] = pc[:, :3].contiguous()
        pc = state["cloud"]
        reg = state["src"] if source_downsampled else state["xyz"]
        if ev: ev[1].record()
        gn = trk._engine(reg.shape[0], gp, cfg.reg_lm_lambda)
        gn.on_knn, gn.on_gn = hooks
        T, cnt, res_cm, its, _, _ = gn.track(reg, T_init, args.reg_iters, early_exit=False)
        gn.on_knn = gn.on_gn = None
        if ev: ev[2].record()
        if args.stages == "all":
            mp.process_frame(pc, None, pose_t, fid)
        if ev: ev[3].record()
        mp.mapping(args.map_iters)
        if ev:
            ev[4].record()
            stage_events.append(ev)
        stats["last"] = (T, cnt, res_cm, its, reg.shape[0], gn)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        frame(False)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        frame(True, hooks=(on_knn, on_gn) if i < ev_frames else (None, None))
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    T, cnt, res_cm, its, n_reg, gn = stats["last"]
    nn_mean = float(gn.nn[:n_reg].float().mean().item())
    names = ("preprocess", "odometry", "map_prep", "mapping")
    stage_ms = {n: round(float(np.mean([e[i].elapsed_time(e[i + 1]) for e in stage_events])), 3) for i, n in enumerate(names)}
    pool_now, new_now, n_src = mp.pool_sample_count, (0 if mp.new_idx is None else int(mp.new_idx.shape[0])), int(state["src"].shape[0])

    # the reference's own odometry workload: register the source-down-sampled subset (reported, not `value`)
    frame(False, source_downsampled=True)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        frame(False, source_downsampled=True)
    barrier()
    elapsed_ds = time.perf_counter() - t0

    # achievable HBM ceiling on this box: a 1 GiB device-to-device copy (read + write), outside the timed regions
    ca = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
    cb = torch.empty_like(ca)
    for _ in range(2):
        cb.copy_(ca)
    torch.cuda.synchronize()
    c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    c0.record()
    for _ in range(5):
        cb.copy_(ca)
    c1.record()
    torch.cuda.synchronize()
    copy_gbs = 5 * 2 * ca.numel() * 4 / (c0.elapsed_time(c1) * 1e-3) / 1e9
    del ca, cb
    knn_ms = float(np.mean([a.elapsed_time(b) for a, b in ev_pairs])) if ev_pairs else float("nan")
    gn_ms = float(np.mean([a.elapsed_time(b) for a, b in gn_pairs])) if gn_pairs else float("nan")
    # fused SDF + Jacobian + GN kernel: decoder flops per query, forward + input Jacobian
    flops_q = 2 * 2 * (11 * H + (L - 1) * H * H + H)
    gn_tflops = flops_q * n_reg / (gn_ms * 1e-3) / 1e12
    # what the matrix cores execute: every fp32 product as six bf16 piece products (mlp_bf3.h), layer 0 padded to K = 16
    split_bf16 = os.environ.get("PIN_MLP", "") != "f32"
    exec_flops_q = 6 * 2 * 2 * (16 * H + (L - 1) * H * H) if split_bf16 else 2 * 2 * (16 * H + (L - 1) * H * H)
    exec_tflops = exec_flops_q * n_reg / (gn_ms * 1e-3) / 1e12
    Kc = int(npts.neighbor_K)
    rho = nn_mean / Kc  # measured fraction of candidate cells holding an accepted neural point
    # algorithmic bytes of one kNN launch (DESIGN.md "kernel: knn_query"): query in + out,
    # one 4-byte slot per candidate cell, one 16-byte position per occupied cell, kNN record out
    bytes_q = 12 + 12 + 4 * Kc + 16 * rho * Kc + 16 * k + 4
    achieved = bytes_q * n_reg / (knn_ms * 1e-3) / 1e9
    pmc_data = {}
    pmc = os.path.join(ROOT, "profiles", "r01_pmc.json")
    if os.path.exists(pmc):
        try:
            pmc_data = json.load(open(pmc))
        except Exception:
            pmc_data = {}

    frames_per_s = world * args.steps / elapsed
    ms_step = 1e3 * elapsed / args.steps
    out = {
        "metric": "slam_frames_per_sec", "value": round(frames_per_s, 3), "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {wl['desc']}; frame = preprocess + {args.reg_iters} GN iterations over the "
                               f"whole cropped scan (no early exit) + process_frame (7 samples/ray, map update, pool "
                               f"filter) + {args.map_iters} mapping iterations of batch {args.bs} (+{(args.bs + 9) // 10}x6 Eikonal)"
                               if args.stages == "all" else
                               f"{args.workload}: {wl['desc']}; frame = {args.reg_iters} GN iterations (no early exit) + "
                               f"{args.map_iters} mapping iterations of batch {args.bs}",
                   "neural_points": P, "scan_points": args.scan, "registered_points": n_reg, "knn_k": k,
                   "candidate_cells": Kc, "decoder": f"{L}x{H}", "occupancy_rho": round(rho, 4),
                   "pool_samples": pool_now, "new_samples": new_now, "brick_cache": npts._bricks is not None,
                   "stages": args.stages,
                   "parallelism": "1 GPU" if world == 1 else
                                  (f"preprocess/odometry/map-prep replicas x{world}, mapper dp{world} (global batch {args.bs * world}, "
                                   f"RCCL all-reduce of decoder + feature gradients per iteration)" if mapper_dp else
                                   f"{world} independent replicas (one frame stream per GPU, no data-path collective; "
                                   f"--parallel dp selects the data-parallel mapper)")},
        "stage_ms_per_frame": stage_ms,
        "mapper_samples_per_sec": round(world * args.bs * args.map_iters / (1e-3 * stage_ms["mapping"]), 1),
        "frames_per_sec_source_downsampled": round(world * args.steps / elapsed_ds, 3),
        "source_points": n_src,
        "gn_valid_points": int(cnt), "gn_residual_cm": round(float(res_cm), 4),
        "roofline": {"kernel": "gn_accumulate_quad_kernel", "bound": "mfma", "achieved": round(gn_tflops, 2),
                     "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(gn_tflops / FP32_PEAK_TFLOPS, 4),
                     "traffic": pmc_data.get("gn_hbm_bytes_per_launch"), "avg_launch_ms": round(gn_ms, 4),
                     "launches": len(gn_pairs), "algorithmic_flops_per_query": flops_q,
                     "share_of_frame": round(gn_ms * args.reg_iters / ms_step, 3),
                     "arithmetic": ("fp32 factors split exactly into 3 bf16 pieces, 6 piece products per fp32 product on "
                                    "v_mfma_f32_16x16x32_bf16, fp32 accumulate; `achieved`/`peak` are fp32-equivalent")
                                   if split_bf16 else "v_mfma_f32_16x16x4_f32",
                     "limiter": "vector-ALU + MFMA issue, additive on gfx950 (PMC: ~16.7 M vector instructions = 27 us and "
                                "13 us of MFMA per launch and SIMD, profiles/r01_pmc.json; scripts/mfma_valu_overlap.hip)",
                     "executed": {"flops_per_query": exec_flops_q, "tflops": round(exec_tflops, 1),
                                  "peak": BF16_PEAK_TFLOPS if split_bf16 else FP32_PEAK_TFLOPS,
                                  "frac": round(exec_tflops / (BF16_PEAK_TFLOPS if split_bf16 else FP32_PEAK_TFLOPS), 4)}},
        "roofline_knn": {"kernel": "knn_brick_kernel" if npts._bricks is not None else "knn_query_kernel", "bound": "hbm",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_data.get("knn_brick_hbm_bytes_per_launch"),
                         "avg_launch_ms": round(knn_ms, 4), "launches": len(ev_pairs),
                         "algorithmic_bytes_per_query": round(bytes_q, 1),
                         "measured_copy_gbs": round(copy_gbs, 1),
                         "limiter": "vector-ALU instruction count (PMC: ~17.3 M vector instructions = 28 us per launch and "
                                    "SIMD, profiles/r01_pmc.json), not HBM: fabric traffic is 0.43x the algorithmic bytes",
                         "share_of_frame": round(knn_ms * args.reg_iters / ms_step, 3)},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(m, cfg, scan_np, raw.cpu().numpy(), raw_ts.cpu().numpy(), pool_c, pool_l,
                                           m.features, dec.flat_params().cpu().numpy(), H, L, k,
                                           dec.sdf_scale, args)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def cpu_baseline(m, cfg, scan, raw, raw_ts, pool_c, pool_l, feats, dec, H, L, k, sdf_scale, args):
    """The numpy oracle (a port of the reference's torch-CPU chain) timed on a bounded sample of
    the same workload on this host; odometry and mapping are extrapolated linearly in the query
    count, the preprocess / map-prep stages are timed at full size (they are cheap)."""
    from oracle import pin_oracle as O
    n_s, bs_s = 20000, 4096
    # the oracle works on the synthetic map arrays directly (same voxels / hash as the device map)
    params = O.unpack_decoder(dec, 11, H, L)
    dx, mv = O.search_neighborhood(2, 0.5, m.resolution)
    q = scan[:n_s]
    table64 = m.table.astype(np.int64)

    def reg_step():
        s = O.radius_search(q, table64, m.positions, m.resolution, dx, mv)
        sdf, grad, std, nn, _ = O.query_sdf(q, s, feats, m.positions, params, sdf_scale, k, dtype=np.float32)
        O.registration_step(q, sdf, grad, std, nn, valid_nn_k=k)

    def train_iter():
        def searcher(p):
            s = O.radius_search(p, table64, m.positions, m.resolution, dx, mv)
            return O.query_feature(p, s, feats, m.positions, None, k, weighted_first=False)
        O.train_step(pool_c[:bs_s], pool_l[:bs_s], np.ones(bs_s, np.float32), searcher, feats, m.positions, dec,
                     (11, H, L), sdf_scale, k, dec=10, eps=0.08, dtype=np.float32)

    def prep_and_map_prep():
        i1 = O.voxel_down_sample(raw[:, :3], cfg.vox_down_m)
        pc, ts = raw[i1], raw_ts[i1]
        mk = O.crop_frame_mask(pc, cfg.min_z, cfg.max_z, cfg.min_range, cfg.max_range)
        pc, ts = pc[mk], ts[mk]
        i2 = O.voxel_down_sample(pc[:, :3], cfg.source_vox_down_m)
        T = np.eye(4); T[0, 3] = 0.5
        O.deskewing(pc[i2][:, :3], ts[i2], T)
        n = len(pc)
        g = np.random.default_rng(0)
        coord, label, _, w = O.sample_rays(pc[:, :3], None, g.standard_normal(3 * n, dtype=np.float32),
                                           g.random(2 * n, dtype=np.float32), g.random(n, dtype=np.float32),
                                           surface_range=cfg.surface_sample_range_m, surface_n=3, front_n=2, behind_n=1,
                                           free_begin_ratio=cfg.free_sample_begin_ratio, free_end_dist=cfg.free_sample_end_dist_m,
                                           max_range=cfg.max_range)
        glob = np.concatenate([pool_c, coord])
        O.pool_filter_mask(glob, np.zeros(3), cfg.window_radius)
        st = dict(table=table64.copy(), positions=m.positions, ts_create=np.zeros(len(m.positions), np.int32),
                  ts_update=np.zeros(len(m.positions), np.int32))
        O.map_update(st, coord[np.abs(label) < 0.125], 1, m.resolution, temporal=False)
        O.query_certainty(coord, table64, m.positions, np.zeros(len(m.positions), np.float32), m.resolution)

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
    t0 = time.perf_counter()
    prep_and_map_prep()
    t_prep = time.perf_counter() - t0
    limiter.restore_original_limits()
    frame_s = args.reg_iters * t_reg * (args.scan / n_s) + args.map_iters * t_tr * (args.bs / bs_s)
    if args.stages == "all":
        frame_s += t_prep
    return {"value": round(1.0 / frame_s, 5), "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"numpy oracle: {reps} registration steps on {n_s} scan points ({t_reg*1e3:.0f} ms each) and "
                      f"{reps2} mapping iterations of batch {bs_s} ({t_tr*1e3:.0f} ms each), extrapolated linearly to "
                      f"{args.reg_iters}x{args.scan} + {args.map_iters}x{args.bs}; preprocess + map-prep stages once at "
                      f"full size ({t_prep:.1f} s); one host thread (host has {os.cpu_count()} cores)",
            "registration_queries_per_sec": round(n_s / t_reg, 1), "mapper_samples_per_sec": round(bs_s / t_tr, 1)}


if __name__ == "__main__":
    main()
