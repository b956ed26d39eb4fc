#!/usr/bin/env python3
"""bench.py -- PIN-SLAM per-frame pipeline on MI355X: SLAM frames/s (+ mapper samples/s).

One "step" = one SLAM frame, the reference's own frame definition (pin_slam.py:500-502:
preprocess + odometry + mapping preparation + mapping), on the synthetic workload of SURVEY.md
8(d), config C3 (100k-point scan, ~2.2M neural points, kNN=8, Kc=81, decoder 4x64):
  * preprocess   : SLAMDataset.preprocess_frame data path -- voxel down-sampling (vox_down_m),
                   crop_frame, source down-sampling (source_vox_down_m), deskewing;
  * odometry     : Tracker.tracking = `reg_iters` Gauss-Newton iterations (reference default 50,
                   NO early exit => worst case) registering the WHOLE cropped scan (~100k points, put in
                   Morton order once per frame; the reference registers only the source-down-sampled
                   subset -- that cheaper variant is reported beside it as frames_per_sec_source_downsampled);
  * map prep     : Mapper.process_frame -- 7 samples per ray into the pool, NeuralPoints.update +
                   reset_local_map (+ brick cache), pool window / capacity filter over ~2.7M
                   samples, query_certainty, new-sample index;
  * mapping      : Mapper.mapping = 12 iterations, batch 16384 (+ 6*1639 Eikonal queries), BCE +
                   Eikonal, backward to features and decoder (one fused tile kernel + a streamed MFMA weight
                   gradient per iteration; ONE gather and ONE kNN launch for all twelve iterations -- their
                   inputs do not depend on the training), Adam (exact lazy form: only the rows an iteration
                   reads are visited, bit-identical to the dense step; the decoder's step keeps its staged
                   image current).
Everything runs through the drop-in classes (pin_slam_amd.dropin) on libpinhip.  Inputs (raw
scan, timestamps) are resident in HBM before the timed region.

N > 1 (one rank per GPU; `python bench.py --gpus N` starts its own ranks through torch.distributed.run when it
is not already running under one) measures what the north star asks for at 2/4/8 GPUs: the
data-parallel MAPPER (config C4).  A step = one Mapper.mapping call of `--map-iters` iterations on
a GLOBAL batch of 2^20 samples (`--global-bs`).  Default `--dp-mode spatial` (pin_slam_amd.dp): every
rank trains on the samples of each drawn batch that lie in its k-d box of the voxel grid, lazy exact
Adam on the rows it owns, ONE RCCL all-reduce (through the C ABI, pin_allreduce_f32) of the compact
buffer [decoder grads | halo-row grads] per iteration, the owned rows published once per call;
`--dp-mode dense`: contiguous index shards, all-reduce of the whole [decoder | feature] gradient
(71 MB) per iteration, replicated dense Adam.  certainty / ts side effects are merged once per call.
metric = mapper_samples_per_sec, value = steps * iters * 2^20 / time (strong scaling: the global batch
is fixed).  The N = 1 line carries the same workload on one GPU in `c4_single_gpu`, the N = 1 point of
that curve, and `c4_per_rank_emulated`: single ranks of 2 / 4 / 8-rank jobs run alone on the one GPU
(their share of the work, identity exchange) -- measured per-rank times, with the exchange added from a
stated model.  Registration is single-GPU by nature (sequential GN iterations); `--parallel replicas`
runs N independent frame streams instead (no data-path collective; value = N * frames / time).

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

LIDAR_CFG = dict(voxel_size_m=0.4, search_alpha=0.5, num_nei_cells=2, query_nn_k=8, max_range=80.0, local_map_radius=82.0,
                 window_radius=80.0, vox_down_m=0.08, source_vox_down_m=0.8, min_range=2.5, min_z=-5.0, max_z=80.0, deskew=True)
WORKLOADS = {
    # BASELINE.json configs 2 / 3: LiDAR scan on wavy sheets 0.8 m apart in an 80 m disc
    "c3": dict(layers=16, hidden=64, levels=4, scan=100_000, cfg=LIDAR_CFG, map=dict(),
               desc="100k-pt scan, ~2.2M neural points, kNN=8, Kc=81, decoder 4x64"),
    "c2": dict(layers=4, hidden=32, levels=2, scan=100_000, cfg=LIDAR_CFG, map=dict(),
               desc="100k-pt scan, ~0.56M neural points, kNN=8, Kc=81, decoder 2x32"),
    # config/lidar_slam/run_kitti.yaml style: per-neighbour decoding (weighted_first False: the decoder runs k times per
    # query and the spread of the k predictions gates the registration), class-default 1x64 decoder, k = 6, Kc = 33
    "kitti": dict(layers=16, hidden=64, levels=1, scan=100_000, map=dict(),
                  cfg=dict(LIDAR_CFG, query_nn_k=6, search_alpha=0.2, weighted_first=False),
                  desc="100k-pt scan, ~2.2M neural points, weighted_first=False (per-neighbour decoding), kNN=6, Kc=33, "
                       "decoder 1x64"),
    # BASELINE.json config 5 (config/rgbd_slam/run_replica.yaml): RGB-D frames, 5 cm voxels, colour + SDF decoders (1x64, the
    # class defaults), k = 6, Kc = 33 (search_alpha 0.2), photometric registration, colour L1 in mapping; sheets 0.1 m apart
    # in a 10 m room until ~5 M neural points
    "c5": dict(layers=128, hidden=64, levels=1, scan=300_000, color=True,
               map=dict(radius=5.64, raw_per_layer=200_000, sheets=(-6.4, 0.1, 0.03, 2.0)), scan_noise=0.005, pool_sigma=0.03,
               cfg=dict(voxel_size_m=0.05, search_alpha=0.2, num_nei_cells=2, query_nn_k=6, max_range=10.0, local_map_radius=12.0,
                        window_radius=10.0, vox_down_m=0.02, source_vox_down_m=0.06, min_range=0.05, min_z=-10.0, max_z=80.0,
                        deskew=False, color_on=True, color_channel=3, photometric_loss_on=True, photometric_loss_weight=0.01,
                        surface_sample_range_m=0.03, free_sample_end_dist_m=0.1, free_front_n=1, sigma_sigmoid_m=0.01,
                        weight_e=0.2, reg_min_grad_norm=0.4, reg_max_grad_norm=2.5, reg_GM_grad=0.3, reg_GM_dist_m=0.05,
                        eigenvalue_check=False),
               desc="300k-pt RGB-D frame, ~5.3M neural points, 5 cm voxels, kNN=6, Kc=33, SDF + colour decoders 1x64, "
                    "photometric registration, colour L1 in mapping"),
}
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 vector = fp32 MFMA rate
F16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense fp16 / bf16 MFMA


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c3", choices=list(WORKLOADS))
    ap.add_argument("--reg-iters", type=int, default=50)
    ap.add_argument("--map-iters", type=int, default=12)
    ap.add_argument("--bs", type=int, default=16384)
    ap.add_argument("--scan", type=int, default=None, help="points in the raw scan (default: the workload's, 100k / 300k)")
    ap.add_argument("--pool", type=int, default=2_000_000, help="samples in the pool (= pool_capacity)")
    ap.add_argument("--pretrain-iters", type=int, default=300)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle check of the timed kernels on the timed workload")
    ap.add_argument("--parallel", default="dp", choices=["dp", "replicas"],
                    help="N > 1: 'dp' (default) = the data-parallel mapper of config C4: a global batch of --global-bs "
                         "samples sharded over the ranks, one RCCL all-reduce of [decoder | feature] gradients per "
                         "iteration (SURVEY 8e); 'replicas' = N independent frame streams, no data-path collective")
    ap.add_argument("--global-bs", type=int, default=1 << 20, help="global mapper batch of the dp / C4 measurement")
    ap.add_argument("--dp-mode", default="spatial", choices=["spatial", "dense"],
                    help="how the mapper batch is cut over the ranks: spatial = k-d boxes of the voxel grid, halo-row exchange "
                         "(pin_slam_amd.dp); dense = contiguous index shards, all-reduce of the whole gradient table")
    ap.add_argument("--dp-transport", default="rccl", choices=["rccl", "auto", "torch", "host"],
                    help="rccl = the product transport (one rank per GPU; RCCL through the C ABI, self-tested at start-up): if ANY "
                         "rank fails to bring it up every rank exits with status 3 -- this command never times another transport "
                         "in its place.  auto = rccl, and if it fails all ranks use torch.distributed's communicator together "
                         "(the line's allreduce.transport says so).  "
                         "torch = torch.distributed's RCCL communicator.  host = TEST mode for a single-GPU box: the ranks share "
                         "cuda:0 and exchange through pinned host buffers over gloo (collective.HostStagedComm) -- the N > 1 code "
                         "path end to end, not a measurement")
    ap.add_argument("--metric", default="frames", choices=["frames", "mapper"],
                    help="N = 1 only: frames (default) = slam_frames_per_sec, the headline; mapper = the N = 1 point of the `--gpus N` "
                         "curve: mapper_samples_per_sec of Mapper.mapping at --global-bs on this one GPU, the same line format, "
                         "metric, step definition and flags as N > 1 (README: `--gpus 1 --metric mapper`, `--gpus 2`, `--gpus 4`, "
                         "`--gpus 8` form one strong-scaling curve)")
    ap.add_argument("--dp-emulate", default="2,4,8", help="N = 1: world sizes whose single ranks are run alone on this GPU "
                                                          "(c4_per_rank_emulated; empty = skip)")
    ap.add_argument("--emulate-rank", default="", help="W:r -- profiling aid: ONLY rank r of a W-rank data-parallel mapper, alone on "
                                                        "this GPU with the identity exchange (1:0 = the single-GPU mapper); --steps "
                                                        "calls of --map-iters iterations at --global-bs, one JSON line, exit")
    ap.add_argument("--coherent-probe", action="store_true",
                    help="one extra registration, synchronised per iteration: pose step and share of the queries on the coherent "
                         "search path per Gauss-Newton iteration (knn_coherent_probe in the line)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="start the ranks, rendez-vous over gloo on the CPU, print one JSON line and exit (no GPU needed): "
                         "checks the launcher of `--gpus N`")
    ap.add_argument("--c4-iters", type=int, default=12, help="N = 1: iterations per timed Mapper.mapping call of the "
                                                              "single-GPU C4 leg and of the emulated ranks (0 = skip them)")
    ap.add_argument("--moving-steps", type=int, default=10,
                    help="N = 1: frames of the moving-sensor leg reported beside the static workload (the sensor advances "
                         "--moving-m metres per frame along x, every frame has its own scan, NeuralPoints.update appends the "
                         "newly seen surface and the brick cache changes; 0 = skip)")
    ap.add_argument("--moving-m", type=float, default=0.5)
    ap.add_argument("--mesher-queries", type=int, default=10_000_000,
                    help="N = 1: grid points of the Mesher.query_points leg (forward-only SDF + mask over a dense grid, "
                         "mesher.py:40-164; 0 = skip)")
    ap.add_argument("--semantic-leg", type=int, default=1,
                    help="N = 1: 1 = report the semantic head beside the frame (run_demo_sem.yaml's branch, not part of `value`): "
                         "Mapper.mapping with the 21-head semantic decoder and the NLL term on the timed map, and query_sem over "
                         "the timed scan; 0 = skip")
    ap.add_argument("--skip-downsampled", action="store_true",
                    help="skip the second timed leg (registering the source-down-sampled subset): profiles of this "
                         "command then hold one launch shape per tracker kernel")
    ap.add_argument("--force-dp", action="store_true",
                    help="run the data-parallel mapper measurement with whatever world size there is (a one-rank RCCL "
                         "communicator on a single-GPU box: exercises the N > 1 code path end to end)")
    ap.add_argument("--events", default="all", choices=["none", "knn", "all"],
                    help="HIP events around the tracker's kNN / GN launches inside the timed region")
    ap.add_argument("--preprocess-stream", default="side", choices=["side", "main"],
                    help="side (default) = the scan chain of frame f+1 (down-sampling, crop, deskew: reads the raw scan and the last "
                         "odometry step, never the map) is queued on its own stream and runs beside the tail of frame f's "
                         "Mapper.mapping, as a loader thread would queue it; main = on the main stream behind the mapping (r01-r05)")
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


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks ourselves (one per GPU, rendez-vous on
    127.0.0.1) and hand their exit code back.  Rank 0 prints the JSON line on the inherited stdout."""
    import socket
    import subprocess
    if not args.dry_launch and args.dp_transport != "host":
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: this machine shows {have} GPU(s); one rank per GPU is required "
                             f"(RCCL refuses two ranks on one device)")
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(cmd, env=env).returncode


def bring_up_transport(args, rank, world, device="cuda"):
    """The data-parallel mapper's transport as --dp-transport asks for it.  rccl is STRICT: a failure on any rank ends every
    rank with exit status 3 and a JSON error on stderr -- the N > 1 line of this command is never a measurement of another
    transport (a SCALE run on torch.distributed would still have "passed")."""
    from pin_slam_amd import collective
    strict = args.dp_transport == "rccl"
    try:
        return collective.make_comm(rank, world, "rccl" if args.dp_transport == "auto" else args.dp_transport, device=device,
                                    fallback=not strict)
    except collective.TransportError as e:
        if rank == 0:
            print(json.dumps({"error": "dp transport", "requested": args.dp_transport, "n_gpus": world, "detail": str(e)}),
                  file=sys.stderr, flush=True)
        sys.stderr.flush()
        if getattr(e, "abandoned_thread", False):
            os._exit(3)
        raise SystemExit(3)


def dry_launch(rank, world):
    """The launcher's self-test: the ranks meet over gloo on the CPU and add up their ranks."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"dry_launch": True, "ranks": world, "rank_sum": float(t.item()),
                          "expected": world * (world + 1) / 2.0}), flush=True)
    dist.destroy_process_group()


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args))
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE is {world}")
    if args.dry_launch:
        return dry_launch(rank, world)
    shared_gpu = args.dp_transport == "host"
    if shared_gpu:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    from pin_slam_amd import preprocess, synth
    from pin_slam_amd.config import PinConfig
    from pin_slam_amd.dropin.model.decoder import Decoder
    from pin_slam_amd.dropin.model.neural_points import NeuralPoints
    from pin_slam_amd.dropin.utils.mapper import Mapper
    from pin_slam_amd.dropin.utils.tracker import Tracker

    wl = WORKLOADS[args.workload]
    H, L, k = wl["hidden"], wl["levels"], int(wl["cfg"]["query_nn_k"])
    colour = bool(wl.get("color", False))
    if args.scan is None:
        args.scan = wl["scan"]
    mapper_dp = (world > 1 or args.force_dp) and args.parallel == "dp"
    res = wl["cfg"]["voxel_size_m"]
    n_frames = args.warmup + 2 * args.steps + 4 + args.moving_steps + 2
    cfg = PinConfig(buffer_size=int(5e7), feature_std=0.1, bs=args.global_bs if mapper_dp else args.bs, iters=args.map_iters,
                    local_map_travel_dist_ratio=5.0, pool_capacity=args.pool, pool_filter_freq=1, bs_new_sample=2048,
                    geo_mlp_level=L, geo_mlp_hidden_dim=H, color_mlp_level=L, color_mlp_hidden_dim=H,
                    reg_iter_n=args.reg_iters, **wl["cfg"])
    torch.manual_seed(42)  # identical on every rank: the ranks must keep identical maps and draw identical batches

    # ---------------- synthetic map / scan / pool (identical on every rank) ----------------
    m = synth.build_map(layers=wl["layers"], resolution=res, **wl["map"])
    npts = NeuralPoints(cfg)
    npts.travel_dist = torch.zeros(n_frames + 1, dtype=torch.float32, device="cuda")
    npts.update(torch.from_numpy(m.positions).cuda(), torch.zeros(3), torch.eye(3), 0)
    P = npts.count()
    dec = Decoder(cfg, H, L, 1)
    cdec = Decoder(cfg, H, L, 3) if colour else None
    decoders = {"sdf": dec, "semantic": None, "color": cdec}
    ds = Dataset(n_frames + 1)
    mp = Mapper(cfg, ds, npts, decoders)
    single_gpu_c4 = None
    trk = Tracker(cfg, npts, decoders)
    pool_c, pool_l = synth.make_pool(m, n=args.pool, sigma=wl.get("pool_sigma", 0.25))
    crng = np.random.default_rng(9)
    if colour:
        mp.color_pool = torch.from_numpy(crng.random((len(pool_l), 3), dtype=np.float32)).cuda()
    mp.coord_pool = torch.from_numpy(pool_c).cuda()
    mp.global_coord_pool = mp.coord_pool.clone()  # poses are identity in this workload
    mp.sdf_label_pool = torch.from_numpy(pool_l).cuda()
    mp.weight_pool = torch.ones(len(pool_l), dtype=torch.float32, device="cuda")
    mp.time_pool = torch.zeros(len(pool_l), dtype=torch.int32, device="cuda")
    mp.pool_sample_count = len(pool_l)
    mp._pool()          # adopt the tensors into the device pool
    mp._publish_pool()

    def barrier():
        # (this rank's own queue first: the job's process group and the mapper's communicator are two RCCL communicators, and a
        # barrier kernel queued beside a still-running all-reduce of the other one would have the two spin on each other's ranks)
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world > 1:
            import torch.distributed as dist
            t = torch.tensor([x], dtype=torch.float64, device="cpu" if shared_gpu else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return x

    if args.emulate_rank:
        from pin_slam_amd import collective
        W, r = (int(v) for v in args.emulate_rank.split(":"))
        cfg.bs = args.global_bs
        if W > 1:
            mp.dp_rank, mp.dp_world, mp.dp_comm, mp.dp_mode = r, W, collective.NullComm(r, W), args.dp_mode
        for _ in range(max(1, args.warmup)):
            mp.mapping(args.map_iters)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            mp.mapping(args.map_iters)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({"emulated_rank": r, "world": W, "dp_mode": args.dp_mode if W > 1 else None, "global_batch": args.global_bs,
                          "iterations_per_call": args.map_iters, "ms_per_call": round(1e3 * dt / args.steps, 4),
                          "ms_per_iteration": round(1e3 * dt / (args.steps * args.map_iters), 4),
                          "shards": getattr(mp, "dp_stats", None)}), flush=True)
        return
    if args.metric == "mapper" and world == 1 and not mapper_dp:
        # the N = 1 point of the `--gpus N` curve: the same metric, step and line format as bench_dp_mapper
        for _ in range(max(1, args.pretrain_iters // 50)):
            mp.mapping(10)
        print(json.dumps(bench_single_gpu_mapper(args, cfg, mp, npts, wl, P)), flush=True)
        return
    if mapper_dp:
        # the same workload on ONE GPU first, on this box in this run: the N = 1 point the N-rank value is set against
        for _ in range(max(1, args.pretrain_iters // 50)):
            mp.mapping(10)
        if args.c4_iters > 0:
            single_gpu_c4 = c4_single_gpu(args, cfg, mp)
        # RCCL through the C ABI; torch.distributed only carries the ncclUniqueId (and this script's barriers)
        from pin_slam_amd import collective
        comm = bring_up_transport(args, rank, world)
        mp.dp_rank, mp.dp_world, mp.dp_comm, mp.dp_mode = rank, world, comm, args.dp_mode
        out = bench_dp_mapper(args, cfg, mp, npts, wl, rank, world, barrier, max_over_ranks, P, single_gpu_c4)
        mp.dp_comm.close()
        if world > 1:
            import torch.distributed as dist
            dist.destroy_process_group()
        # the JSON line is the LAST thing on stdout: RCCL's version banner sits in the C runtime's buffer until exit
        import ctypes
        ctypes.CDLL(None).fflush(None)
        if rank == 0:
            print(json.dumps(out), flush=True)
        if getattr(comm, "abandoned_thread", False):
            os._exit(0)  # (a bootstrap thread of the transport that was not used is still blocked in the library)
        return

    scan_np = synth.make_scan(m, n=args.scan, seed=1, noise=wl.get("scan_noise", 0.02))
    rng = np.random.default_rng(5)
    # rows of the raw scan: xyz + intensity (LiDAR) or xyz + rgb (RGB-D)
    raw = torch.from_numpy(np.concatenate([scan_np, rng.random((args.scan, 3 if colour else 1), dtype=np.float32)], 1)).cuda()
    raw_ts = torch.from_numpy(rng.random(args.scan, dtype=np.float32)).cuda()
    last_odom = np.eye(4)
    last_odom[:3, 3] = [0.5, 0.0, 0.0]

    # pre-train so that the SDF is a real field (GN accepts points); not timed
    for _ in range(max(1, args.pretrain_iters // 50)):
        mp.mapping(50)
    ang = 0.003
    T_init = np.eye(4)
    T_init[:3, :3] = [[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]]
    T_init[:3, 3] = np.array([0.05, -0.04, 0.02]) * (res / 0.4)  # an eighth of a voxel off
    gp = trk._gn_params(cfg.reg_min_grad_norm, cfg.reg_max_grad_norm, cfg.reg_GM_dist_m, cfg.reg_GM_grad)
    prep = preprocess.ScanPreprocessor(cfg)
    pose_t = torch.eye(4, dtype=torch.float64, device="cuda")

    # HIP events around every GN-accumulate launch (the dominant kernel of the frame) and every kNN
    # launch of the tracker, recorded on the launch stream inside the timed region
    ev_pairs, gn_pairs = [], []
    cur_ev = {}
    # the first timed frames, every EV_STRIDE-th launch of each kernel: an event pair costs ~3 us on the launch stream -- all
    # 2 x 2 x 50 of a frame stretched its odometry from 3.55 to 4.15 ms, i.e. the instrumented frames pulled `value` down by
    # 1.6 % at --steps 20 and by 6 % at --steps 5; 2 x 10 samples per kernel give the same average
    EV_STRIDE = 5
    ev_frames = min(2, args.steps)
    n_ev = ev_frames * args.reg_iters
    evpool = {t: [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_ev)]
              for t in ("k", "g")}  # created (and warmed) outside the timed region
    for t in evpool:
        for a_, b_ in evpool[t]:
            a_.record(); b_.record()
    torch.cuda.synchronize()

    def bracket(store, tag):
        seen = {"n": 0}

        def hook(start):
            if start:
                seen["n"] += 1
                cur_ev[tag] = evpool[tag][len(store)] if seen["n"] % EV_STRIDE == 1 else None
                if cur_ev[tag] is not None:
                    cur_ev[tag][0].record()
            elif cur_ev[tag] is not None:
                cur_ev[tag][1].record()
                store.append(cur_ev[tag])
        return hook

    on_knn = bracket(ev_pairs, "k") if args.events in ("knn", "all") else None
    on_gn = bracket(gn_pairs, "g") if args.events == "all" else None
    stats = {}
    state = {"fid": 1, "cloud": None, "src": None}

    stage_events = []  # per timed frame: 5 events at the stage boundaries (no host sync between the stages)
    # (default priority: a high-priority stream changed nothing for the frame -- 196.8 / 200.9 against 198.4 / 198.2 frames/s, same box --
    # but its existence slowed ONE of the streams the data-parallel mapper's legs create later in this process (one emulated rank
    # per world 30-45 % slower: the runtime maps streams to a handful of hardware queues by priority class))
    side = torch.cuda.Stream(priority=int(os.environ.get("PIN_BENCH_SIDE_PRIORITY", "0"))) if args.preprocess_stream == "side" else None
    side_events = []   # per timed frame: the scan chain's own two events on its stream
    host_marks = []    # per timed frame: host clock at the same boundaries (how long the host takes to ENQUEUE a stage)

    def frame(timed, hooks=(None, None), source_downsampled=False, raw=raw, T_init=T_init, pose_t=pose_t, sink=stage_events):
        fid = state["fid"]
        state["fid"] += 1
        ds.processed_frame = fid
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)] if timed else None
        hm = [time.perf_counter()]
        if ev: ev[0].record()
        if args.stages == "all" or state["cloud"] is None:
            se = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if (ev and side is not None) else None
            if se: se[0].record(side)
            _ = prep(raw, raw_ts, last_odom_tran=last_odom, frame_id=fid, stream=side)
            pc, src = _[0], _[2]
            state["cloud"], state["src"] = pc, src
            if side is not None:
                # ... and what the registration needs of the cloud: its coordinates as a table of their own, Morton-ordered into the
                # tracker's buffer (GNTracker.presort) -- still on the scan's stream, beside the previous frame's mapping
                main_stream = torch.cuda.current_stream()
                with torch.cuda.stream(side):
                    state["xyz"] = pc[:, :3].contiguous()
                    state["xyz"].record_stream(main_stream)
                    if trk._gn is not None and not source_downsampled and os.environ.get("PIN_BENCH_PRESORT", "1") != "0":  # (A/B switch)
                        trk._gn.presort(state["xyz"], stream=side)
                    side_done = torch.cuda.Event()
                    side_done.record(side)
                main_stream.wait_event(side_done)  # (queued behind the previous frame's mapping: the copy is long done by then)
            else:
                state["xyz"] = pc[:, :3].contiguous()
            if se:
                se[1].record(side)
                if sink is stage_events:
                    side_events.append(se)
            state["rgb"], state["src_rgb"] = (pc[:, 3:6].contiguous(), _[3]) if colour else (None, None)
        pc = state["cloud"]
        reg = state["src"] if source_downsampled else state["xyz"]
        hm.append(time.perf_counter())
        if ev: ev[1].record()
        gn = trk._engine(reg.shape[0], gp, cfg.reg_lm_lambda)
        gn.on_knn, gn.on_gn = hooks
        # colour term of the registration (photometric rows / consistency weights, tracker.py:492-518)
        ct, _keep = trk._color_term(state["src_rgb"] if source_downsampled else state["rgb"]) if colour else (None, None)
        T, cnt, res_cm, its, _, _ = gn.track(reg, T_init, args.reg_iters, early_exit=False, color=ct)
        gn.on_knn = gn.on_gn = None
        hm.append(time.perf_counter())
        if ev: ev[2].record()
        if args.stages == "all":
            mp.process_frame(pc, None, pose_t, fid)
        hm.append(time.perf_counter())
        if ev: ev[3].record()
        mp.mapping(args.map_iters)
        hm.append(time.perf_counter())
        if ev:
            ev[4].record()
            sink.append(ev)
            if sink is stage_events:
                host_marks.append(hm)
        stats["last"] = (T, cnt, res_cm, its, reg.shape[0], gn)

    for i in range(args.warmup):
        frame(False)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        frame(True, hooks=(on_knn, on_gn) if i < ev_frames else (None, None))
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    T, cnt, res_cm, its, n_reg, gn = stats["last"]
    nn_mean = float(gn.nn[:n_reg].float().mean().item())
    names = ("preprocess", "odometry", "map_prep", "mapping")
    stage_ms = {n: round(float(np.mean([e[i].elapsed_time(e[i + 1]) for e in stage_events])), 3) for i, n in enumerate(names)}
    host_ms = {n: round(1e3 * float(np.mean([h[i + 1] - h[i] for h in host_marks])), 3) for i, n in enumerate(names)}
    stage_note = None
    if side_events:  # the scan chain ran on its own stream: its time is what ITS events say; the main stream's first stage is the wait for it
        stage_ms["preprocess_main_stream_wait"] = stage_ms["preprocess"]
        stage_ms["preprocess"] = round(float(np.mean([a.elapsed_time(b) for a, b in side_events])), 3)
        stage_note = ("preprocess of frame f+1 runs on its own stream beside the tail of frame f's mapping (--preprocess-stream side): "
                      "its time is measured on that stream and the stages sum to MORE than ms_per_step; "
                      "preprocess_main_stream_wait is what the main stream spends between the frame's start and the registration")
    pool_now, new_now, n_src = mp.pool_sample_count, (0 if mp.new_idx is None else int(mp.new_idx.shape[0])), int(state["src"].shape[0])

    probe = None
    if args.coherent_probe:
        probe = []
        gnp = trk._engine(state["xyz"].shape[0], gp, cfg.reg_lm_lambda)
        gnp.track(state["xyz"], T_init, args.reg_iters, early_exit=False, probe=probe)
        for r in probe:
            r.pop("_T")
    # the reference's own odometry workload: register the source-down-sampled subset (reported, not `value`)
    elapsed_ds = None
    if not args.skip_downsampled:
        frame(False, source_downsampled=True)
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            frame(False, source_downsampled=True)
        barrier()
        elapsed_ds = time.perf_counter() - t0

    # parity is taken on the map as the static workload leaves it; the moving-sensor leg below changes the map, so the device
    # side of the parity check comes first
    parity_in = None
    if rank == 0 and world == 1 and not args.no_parity:
        parity_in = gpu_parity_sample(npts, dec, trk, gp, state["xyz"], k, rgb=state["rgb"], mp=mp, cdec=cdec)

    # the same frame with a MOVING sensor (VERDICT r3 missing #5): the pose advances --moving-m per frame, every frame has its
    # own scan of the surface around the new position, so NeuralPoints.update appends the newly seen surface, the table takes
    # inserts, the local map and the brick cache change -- the static workload above never exercises that inside a timed frame
    moving = None
    if world == 1 and args.moving_steps > 0 and args.stages == "all" and not colour:
        moving = bench_moving(args, wl, m, cfg, npts, mp, frame, barrier, names, P, scan_np)

    mesher_leg = None
    if world == 1 and args.mesher_queries > 0 and not colour:
        mesher_leg = bench_mesher(args, cfg, npts, decoders, nn_mean, int(npts.neighbor_K), k)

    semantic_leg = None
    if world == 1 and args.semantic_leg and not colour and args.stages == "all":
        semantic_leg = bench_semantic(args, cfg, npts, dec, mp, state["xyz"], H, L, k)

    # the GPU side of the parity check (the oracle side runs in cpu_baseline_and_parity, outside every timed region):
    # the benchmarked kernels -- brick kNN + the GN tile kernel -- on a sub-sample of the timed scan, against the
    # map as it stands after the timed frames
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

    # config C4 on ONE GPU (the N = 1 point of the data-parallel mapper curve): Mapper.mapping on a 2^20 batch
    c4 = None
    c4_emul = None
    if world == 1 and args.c4_iters > 0 and not colour:
        c4 = c4_single_gpu(args, cfg, mp, nn_mean, int(npts.neighbor_K), k)
        worlds = [int(w) for w in args.dp_emulate.split(",") if w.strip()]
        if worlds and cfg.weighted_first:
            c4_emul = c4_per_rank_emulated(args, cfg, mp, worlds, c4)

    knn_ms = float(np.mean([a.elapsed_time(b) for a, b in ev_pairs])) if ev_pairs else float("nan")
    gn_ms = float(np.mean([a.elapsed_time(b) for a, b in gn_pairs])) if gn_pairs else float("nan")
    # fused SDF + Jacobian + GN kernel: decoder flops per query, forward + input Jacobian (colour: + the 3-head decoder)
    flops_q = 2 * 2 * (11 * H + (L - 1) * H * H + H) + (2 * 2 * (11 * H + (L - 1) * H * H + 3 * H) if colour else 0)
    if not cfg.weighted_first:
        flops_q *= k  # the decoder runs once per neighbour
    gn_tflops = flops_q * n_reg / (gn_ms * 1e-3) / 1e12
    # what the matrix cores execute: every fp32 product as three fp16 piece products (mlp_h2.h), layer 0 padded to K = 16;
    # the colour / per-neighbour kernel (64 queries per wave) stays on the fp32 MFMA
    split_f16 = os.environ.get("PIN_MLP", "") != "f32" and cfg.weighted_first and (not colour or L <= 2)
    exec_flops_q = (3 if split_f16 else 1) * (2 if colour else 1) * 2 * 2 * (16 * H + (L - 1) * H * H)
    exec_tflops = exec_flops_q * n_reg / (gn_ms * 1e-3) / 1e12
    Kc = int(npts.neighbor_K)
    rho = nn_mean / Kc  # measured fraction of candidate cells holding an accepted neural point
    # algorithmic bytes of one kNN launch (DESIGN.md "kernel: knn_query"): query in + out,
    # one 4-byte slot per candidate cell, one 16-byte position per occupied cell, kNN record out
    bytes_q = 12 + 12 + 4 * Kc + 16 * rho * Kc + 16 * k + 4
    achieved = bytes_q * n_reg / (knn_ms * 1e-3) / 1e9
    # counters of THIS command on this workload (scripts/pmc_bench.sh -> profiles/r03_pmc_<workload>.json): HBM-side bytes per
    # launch, matrix-pipe and vector-ALU utilisation of the two tracker kernels
    pmc_data, pmc_src = {}, None
    for rnd in ("r06", "r05", "r04", "r03"):  # (the newest committed counter passes of this workload)
        path = os.path.join(ROOT, "profiles", f"{rnd}_pmc_{args.workload}.json")
        if os.path.exists(path):
            try:
                pmc_data, pmc_src = json.load(open(path)), f"profiles/{rnd}_pmc_{args.workload}.json"
                break
            except Exception:
                pass
    gnk = pmc_data.get("kernels", {}).get("gn", {})
    # (the tracker's search is knn_brick_listed_kernel since r04 -- candidate lists across the iterations; PIN_KNN_LISTED=0: knn_brick_kernel)
    listed_on = os.environ.get("PIN_KNN_LISTED", "1") != "0" and npts._bricks is not None
    knk = pmc_data.get("kernels", {}).get("knn_brick_listed" if listed_on else "knn_brick", {}) or pmc_data.get("kernels", {}).get("knn_brick", {})

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
                                  f"{world} independent replicas (one frame stream per GPU, no data-path collective; "
                                  f"the default --parallel dp measures the data-parallel mapper instead)"},
        "stage_ms_per_frame": stage_ms, "stage_note": stage_note,
        "host_enqueue_ms_per_frame": host_ms,
        "mapper_samples_per_sec": round(world * args.bs * args.map_iters / (1e-3 * stage_ms["mapping"]), 1),
        "frames_per_sec_source_downsampled": None if elapsed_ds is None else round(world * args.steps / elapsed_ds, 3),
        "source_points": n_src,
        "gn_iterations": int(its), "gn_valid_points": int(cnt), "gn_residual_cm": round(float(res_cm), 4),
        "host_trace_ms": (lambda hc: hc.trace_summary() if hc.TRACE else None)(__import__("pin_slam_amd.hostcache", fromlist=["x"])),
        "knn_coherent_probe": probe,
        "moving_sensor": moving,
        "mesher": mesher_leg,
        "semantic": semantic_leg,
        "c4_single_gpu": c4,
        "c4_per_rank_emulated": c4_emul,
        "roofline": {"kernel": ("gn_accumulate_quad_kernel<COLOR> (SDF + colour decoders on two split-fp16 images, photometric rows)"
                                if cfg.weighted_first and L <= 2 and os.environ.get("PIN_MLP", "") != "f32" else
                                "gn_accumulate_mfma_kernel (SDF + colour decoders, photometric rows; 64 queries per wave)")
                               if colour else ("gn_accumulate_quad_kernel" if cfg.weighted_first else
                                               "gn_accumulate_quad_nwf_kernel (per-neighbour decoding: a column per (query, neighbour) pair)"
                                               if L == 1 else
                                               "gn_accumulate_mfma_kernel (per-neighbour decoding; 64 queries per wave)"),
                     "bound": "valu+latency",
                     "bound_note": "vector-ALU issue and gather latency (PMC: matrix pipes busy mfma_util of the SIMD-cycles, vector "
                                   "ALU valu_active); `achieved` / `peak` state the decoder's algorithmic fp32 flops against the fp32 "
                                   "peak for scale -- the kernel does not sit on the matrix-core roofline",
                     "achieved": round(gn_tflops, 2),
                     "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(gn_tflops / FP32_PEAK_TFLOPS, 4),
                     "traffic": gnk.get("hbm_bytes_per_launch"), "traffic_source": pmc_src,
                     "mfma_util": gnk.get("mfma_util"), "valu_active": gnk.get("valu_active"),
                     "avg_launch_ms": round(gn_ms, 4),
                     "launches": len(gn_pairs), "algorithmic_flops_per_query": flops_q,
                     "share_of_frame": round(gn_ms * args.reg_iters / ms_step, 3),
                     "arithmetic": ("fp32 factors split into 2 fp16 pieces (hi + 2^-11 lo, representation error <= 2^-24), 3 piece "
                                    "products per fp32 product on v_mfma_f32_16x16x32_f16, fp32 accumulate; "
                                    "`achieved`/`peak` are fp32-equivalent")
                                   if split_f16 else "v_mfma_f32_16x16x4_f32",
                     "valu_insts_per_launch": gnk.get("SQ_INSTS_VALU"),
                     "mfma_busy_cycles_per_launch": gnk.get("SQ_VALU_MFMA_BUSY_CYCLES"),
                     "executed": {"flops_per_query": exec_flops_q, "tflops": round(exec_tflops, 1),
                                  "peak": F16_PEAK_TFLOPS if split_f16 else FP32_PEAK_TFLOPS,
                                  "frac": round(exec_tflops / (F16_PEAK_TFLOPS if split_f16 else FP32_PEAK_TFLOPS), 4)}},
        "roofline_knn": {"kernel": ("knn_brick_listed_kernel (candidate lists kept across the GN iterations)" if listed_on else "knn_brick_kernel")
                                   if npts._bricks is not None else "knn_query_kernel",
                         "bound": "valu", "bound_note": "vector-ALU instruction issue (PMC), not HBM: the fabric traffic "
                                                        "is below the algorithmic bytes; the GB/s figure is the "
                                                        "algorithmic rate, stated against HBM for scale only",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": knk.get("hbm_bytes_per_launch"),
                         "traffic_source": pmc_src, "valu_active": knk.get("valu_active"),
                         "avg_launch_ms": round(knn_ms, 4), "launches": len(ev_pairs),
                         "algorithmic_bytes_per_query": round(bytes_q, 1),
                         "valu_insts_per_query": None if not knk.get("SQ_INSTS_VALU") else
                                                 round(64.0 * knk["SQ_INSTS_VALU"] / n_reg, 1),
                         "measured_copy_gbs": round(copy_gbs, 1),
                         "share_of_frame": round(knn_ms * args.reg_iters / ms_step, 3)},
    }
    # cpu_baseline = the REAL reference (PRBonn/PIN_SLAM on torch CPU) timed on the GPU box's host by scripts/ref_cpu_baseline.py
    # (BASELINE.md section 3 protocol: 1 warm-up + median of >= 5, thread count and CPU model in the record).  The reference
    # tree cannot be read by this command (it is not part of the repository), so the record is a committed file; the file
    # says which host it was taken on.  cpu_baseline_port = the one-thread numpy oracle timed live by THIS command.
    for name in ("r04_ref_cpu_baseline.json", "r03_ref_cpu_baseline.json"):
        ref_cpu = os.path.join(ROOT, "profiles", name)
        if os.path.exists(ref_cpu) and args.workload == "c3":
            try:
                r = json.load(open(ref_cpu))
                # the record is a stored measurement: say which host THIS command runs on and whether it is the kind of
                # host the record was taken on (CPU model + logical CPU count), so a stale record shows in the line
                now = _host_now()
                rec = {  # (built whole, assigned once: a failure half-way leaves no half-filled entry behind)
                    "value": r["frames_per_sec_bench_definition"], "unit": "frames/s", "cores": r["torch_threads"],
                    "kind": r.get("kind", "reference-torch-cpu"), "host": r["host"], "source": f"profiles/{name}",
                    "sample": (f"unmodified reference classes on torch CPU, C3 inputs of this bench: median of {r['reps']} x "
                               f"Tracker.registration_step over the {r['scan_points']}-point scan ({r['registration_step_ms']} ms) and "
                               f"of {r['reps']} x Mapper.mapping({r['mapping_iterations']}) ({r['mapping_ms']} ms), 1 warm-up each; "
                               f"frame = {args.reg_iters} registration steps + one mapping call (preprocess / map prep not included, "
                               f"which favours the CPU)"),
                    "registration_queries_per_sec": r["registration_queries_per_sec"],
                    "mapper_samples_per_sec": r["mapper_samples_per_sec"], "tracking_ms": r.get("tracking_ms"),
                    "torch": r.get("torch"), "host_now": now["text"],
                    "host_matches_record": bool(now["model"] and now["model"] in r["host"] and f"{now['cpus']} logical CPUs" in r["host"]),
                    "live": False}
                out["cpu_baseline"] = rec
                break
            except Exception:
                pass
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload == "c3":
        live = _ref_cpu_live(args)
        if live is not None:
            out["cpu_baseline_record"] = out.get("cpu_baseline")
            out["cpu_baseline"] = live
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        port = cpu_baseline(
            m, cfg, scan_np, raw.cpu().numpy(), raw_ts.cpu().numpy(), pool_c, pool_l, m.features,
            dec.flat_params().cpu().numpy(), H, L, k, dec.sdf_scale, args)
        out["cpu_baseline_port"] = port
        out.setdefault("cpu_baseline", port)  # (workloads without a committed reference record: the port is the baseline)
    out["parity"] = None
    if parity_in is not None:
        from oracle import pin_oracle as O  # the checker; never inside a timed region
        out["parity"] = parity_vs_oracle(O, parity_in, H, L, k, dec.sdf_scale)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def bench_semantic(args, cfg, npts, dec, mp, scan_xyz, H, L, k):
    """The semantic head on the timed map (config.semantic_on; csrc/sem.h): Mapper.mapping(--map-iters) with a 21-head semantic
    decoder and the NLL term (labels of the pool samples = the sheet they lie on, 1..20) against the same call without it, and
    Tracker.query_source_points-style label queries over the timed scan.  Reported beside the frame; never part of `value`."""
    import copy
    from pin_slam_amd import ops
    from pin_slam_amd.dropin.model.decoder import Decoder
    from pin_slam_amd.dropin.utils.mapper import Mapper
    cfg2 = copy.copy(cfg)
    cfg2.semantic_on, cfg2.sem_class_count, cfg2.weight_s = True, 20, 1.0
    S = cfg2.sem_class_count + 1
    torch.manual_seed(7)
    sem = Decoder(cfg2, H, L, S)
    mp2 = Mapper(cfg2, mp.dataset, npts, {"sdf": dec, "semantic": sem, "color": None})
    n = int(mp.pool_sample_count)
    for a in ("coord_pool", "global_coord_pool", "sdf_label_pool", "weight_pool", "time_pool"):
        setattr(mp2, a, getattr(mp, a)[:n].clone())
    z = mp2.global_coord_pool[:, 2]
    mp2.sem_label_pool = (1 + torch.clamp(((z + 2.0) / 0.8).round(), 0, 19)).to(torch.int32)  # the sheet index (synth.build_map)
    mp2.sem_label_pool[mp2.weight_pool < 0] = 0  # free-space samples carry label 0
    mp2.pool_sample_count, mp2.new_idx = n, None
    mp2._pool(); mp2._publish_pool()
    mp2.determine_used_pose()

    def timed(mapper, reps=5):
        mapper.mapping(args.map_iters)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            mapper.mapping(args.map_iters)
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / reps

    with_sem = timed(mp2)
    t = mp2._get_trainer()
    loss = float(t.sem_loss.item()) / max(int(t.sem_count.item()), 1) / (6 * args.map_iters)
    cfg3 = copy.copy(cfg2)
    cfg3.semantic_on = False
    mp3 = Mapper(cfg3, mp.dataset, npts, {"sdf": dec, "semantic": None, "color": None})
    for a in ("coord_pool", "global_coord_pool", "sdf_label_pool", "weight_pool", "time_pool"):
        setattr(mp3, a, getattr(mp2, a))
    mp3.pool_sample_count, mp3.new_idx = n, None
    mp3._pool(); mp3._publish_pool()
    mp3.determine_used_pose()
    without = timed(mp3)
    q = scan_xyz.contiguous()
    nbr, nn, _ = npts.knn(q, True)
    fsem = npts.field_state(sem, query_locally=True)
    ops.sem_query(fsem, q, nbr, nn, S)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        lab, _ = ops.sem_query(fsem, q, nbr, nn, S)
    e1.record()
    torch.cuda.synchronize()
    q_ms = e0.elapsed_time(e1) / 10
    truth = (1 + torch.clamp(((q[:, 2] + 2.0) / 0.8).round(), 0, 19)).to(torch.int32)
    return {"what": "config.semantic_on on the timed map: 21-head semantic decoder over the geometry features, NLL on the labelled pool "
                    "samples (label = sheet index), csrc/sem.h; beside the frame, not in `value`",
            "mapping_ms_per_call_with_semantic": round(with_sem, 3), "mapping_ms_per_call_without": round(without, 3),
            "semantic_term_us_per_iteration": round(1e3 * (with_sem - without) / args.map_iters, 1),
            "mean_nll_over_the_timed_calls": round(loss, 4),
            "sem_query_ms": round(q_ms, 4), "sem_query_points": int(q.shape[0]),
            "sem_query_points_per_sec": round(q.shape[0] / (q_ms * 1e-3), 1),
            "label_accuracy_on_the_scan_after_the_timed_calls": round(float((lab == truth).float().mean().item()), 4)}


def bench_mesher(args, cfg, npts, decoders, nn_mean, Kc, k):
    """Mesher.query_points (utils/mesher.py:40-164) over a dense regular grid: the forward-only bulk query of the reconstruction
    step (SDF + marching-cubes mask, global map, batches of config.infer_bs), through the drop-in class.  Two numbers: the call
    as the reference defines it (numpy arrays come back: the device -> host copy of the results is inside) and the device part
    alone (search + decode launches between two events)."""
    from pin_slam_amd.dropin.utils.mesher import Mesher
    n = int(args.mesher_queries)
    step = 0.1 * cfg.voxel_size_m / 0.4
    nz = 64
    side = int(np.ceil(np.sqrt(n / nz)))
    ax = (torch.arange(side, device="cuda", dtype=torch.float32) - side / 2) * step
    az = torch.arange(nz, device="cuda", dtype=torch.float32) * step + (-2.0 + 0.8 * 6)  # a slab around sheets 6 .. 14
    coord = torch.stack(torch.meshgrid(ax, ax, az, indexing="ij"), -1).reshape(-1, 3)[:n].contiguous()
    n = coord.shape[0]
    ms = Mesher(cfg, npts, decoders)
    bs = int(cfg.infer_bs)
    keep, ms.global_bricks_min_queries = ms.global_bricks_min_queries, (0 if n >= ms.global_bricks_min_queries else ms.global_bricks_min_queries)
    ms.query_points(coord[:bs], bs, True, False, False, True, False, out_torch=True)  # warm-up (launch shapes, workspaces, the brick cache's buffers)
    ms.global_bricks_min_queries = keep
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sdf, _, _, mask = ms.query_points(coord, bs, True, False, False, True, False, out_torch=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # the device part alone: the same launches without the result copy
    fs = npts.field_state(decoders["sdf"], query_locally=False)
    fs.stage_decoder()
    from pin_slam_amd import ops
    gb = getattr(ms, "_global_bricks", None) if n >= ms.global_bricks_min_queries else None  # (the call's brick cache over the global map)
    st_g = npts.search_state()

    def search(q):
        if gb is not None:
            return ops.knn_query(st_g, q, int(cfg.query_nn_k), time_filtering=False, local=False, bricks=gb)
        return npts.knn(q, False)

    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    if gb is not None:
        gb.build(st_g, time_filtering=False, local=False)  # (inside the device time, as inside the call)
    for a in range(0, n, bs):
        q = coord[a:a + bs]
        nbr, nn, _ = search(q)
        ops.sdf_query(fs, q, nbr, nn, grad=False, std=False, certainty=False)
    e1.record()
    torch.cuda.synchronize()
    dev_ms = e0.elapsed_time(e1)
    # ... and its two halves (a second pass with an event between the search and the decode launch of every batch)
    evs = []
    for a in range(0, n, bs):
        q = coord[a:a + bs]
        ea, eb, ec = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        ea.record()
        nbr, nn, _ = search(q)
        eb.record()
        ops.sdf_query(fs, q, nbr, nn, grad=False, std=False, certainty=False)
        ec.record()
        evs.append((ea, eb, ec))
    torch.cuda.synchronize()
    search_ms = sum(a_.elapsed_time(b_) for a_, b_, _ in evs)
    decode_ms = sum(b_.elapsed_time(c_) for _, b_, c_ in evs)
    rho = nn_mean / Kc
    bytes_q = 12 + 4 * Kc + 16 * rho * Kc + 16 * k + 4 + 36 * k + 4  # search (as roofline_knn) + k feature rows + the SDF out
    # HBM-side bytes per query from the counter passes of this leg (scripts/pmc_bench.sh mesher), when they are committed
    traffic = traffic_src = pmc_note = None
    pm = os.path.join(ROOT, "profiles", "r04_pmc_mesher.json")
    if os.path.exists(pm):
        try:
            ks = json.load(open(pm))["kernels"]
            per_batch = ks["knn_brick"]["hbm_bytes_per_launch"] + ks["sdf_query_quad"]["hbm_bytes_per_launch"]
            traffic, traffic_src = round(per_batch / float(bs), 1), "profiles/r04_pmc_mesher.json"
            pmc_note = {"search_valu_active": ks["knn_brick"].get("valu_active"), "decode_valu_active": ks["sdf_query_quad"].get("valu_active"),
                        "decode_mfma_util": ks["sdf_query_quad"].get("mfma_util")}
        except Exception:
            pass
    return {"queries": n, "grid_step_m": round(step, 4), "batch": bs, "ms_call": round(1e3 * dt, 2),
            "queries_per_sec_call": round(n / dt, 1), "ms_device": round(dev_ms, 2), "queries_per_sec_device": round(n / (dev_ms * 1e-3), 1),
            "ms_search": round(search_ms, 2), "ms_decode": round(decode_ms, 2),
            "valid_share": round(float(mask.float().mean().item()), 4),
            "search": "brick cache over the global map, built per call" if gb is not None else "direct probe of the global table",
            "roofline": {"kernel": ("knn_brick_kernel (global brick cache, built inside the timed region)" if gb is not None else "knn_query_kernel (direct probe of the global table)") + " + sdf_query_quad_kernel (forward only, 4 lanes per query)", "bound": "valu" if pmc_note else "hbm (random 4 / 16 / 32-byte accesses)",
                         "bound_note": ("vector-ALU instruction issue, not HBM: a regular grid's queries share their bricks and rows, the counted "
                                        "HBM-side traffic is a fraction of the algorithmic bytes (PMC of this leg: search valu_active, decode "
                                        "valu_active / mfma_util below); the GB/s figure is the algorithmic rate, stated against HBM for scale only")
                         if pmc_note else None, "pmc": pmc_note,
                         "achieved": round(bytes_q * n / (dev_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(bytes_q * n / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_query": round(bytes_q, 1),
                         "traffic": traffic, "traffic_unit": "HBM-side bytes per query (search + decode launches of one batch)", "traffic_source": traffic_src},
            "note": "ms_call = Mesher.query_points as the reference defines it (results returned on the host); ms_device = its "
                    "search + decode launches alone"}


def bench_moving(args, wl, m, cfg, npts, mp, frame, barrier, names, P0, scan0):
    """`--moving-steps` frames of the same pipeline with the sensor advancing `--moving-m` metres per frame along +x: scan f is
    taken on the map's middle sheet in a disc around the sensor's position (synth.disc_points, so it reaches past the rim of
    the pre-built map on the leading side), handed over in the sensor frame; registration starts from the true pose perturbed
    as in the static workload; Mapper.process_frame gets the true pose, so NeuralPoints.update appends the surface seen for
    the first time and reset_local_map + the brick build follow the sensor.  Scans are generated and uploaded before the
    timed region."""
    from pin_slam_amd import hostcache, synth
    nf = args.moving_steps + 2
    rng = np.random.default_rng(77)
    res = wl["cfg"]["voxel_size_m"]
    ang = 0.003
    dT = np.eye(4)
    dT[:3, :3] = [[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]]
    dT[:3, 3] = np.array([0.05, -0.04, 0.02]) * (res / 0.4)
    raws, poses, inits = [], [], []
    for f in range(nf):
        c = np.array([args.moving_m * (f + 1), 0.0, 0.0])
        pts, _ = synth.disc_points(rng, args.scan, m.radius * 0.95, [m.layers // 2], center=(c[0], c[1]), sheets=m.sheets)
        pts = (pts + wl.get("scan_noise", 0.02) * rng.standard_normal(pts.shape)).astype(np.float32)
        local = (pts - c.astype(np.float32)).astype(np.float32)  # sensor frame (the pose is a pure translation)
        raws.append(torch.from_numpy(np.concatenate([local, rng.random((args.scan, 1), dtype=np.float32)], 1)).cuda())
        T = np.eye(4)
        T[:3, 3] = c
        poses.append(T)
        inits.append(T @ dT)
    torch.cuda.synchronize()
    sink = []
    n_before = npts.count()

    def run(f, timed):
        pt = torch.tensor(poses[f], dtype=torch.float64, device="cuda")
        hostcache.remember(pt, poses[f])  # (as the tracker does for the pose it hands to process_frame)
        frame(timed, raw=raws[f], T_init=inits[f], pose_t=pt, sink=sink)

    for f in range(2):
        run(f, False)
    barrier()
    grown_warm = npts.count() - n_before
    t0 = time.perf_counter()
    for f in range(2, nf):
        run(f, True)
    barrier()
    dt = time.perf_counter() - t0
    steps = nf - 2
    stage = {n: round(float(np.mean([e[i].elapsed_time(e[i + 1]) for e in sink])), 3) for i, n in enumerate(names)}
    worst = {n: round(float(np.max([e[i].elapsed_time(e[i + 1]) for e in sink])), 3) for i, n in enumerate(names)}
    return {"frames_per_sec": round(steps / dt, 3), "ms_per_frame": round(1e3 * dt / steps, 3), "steps": steps,
            "metres_per_frame": args.moving_m, "stage_ms_per_frame": stage, "stage_ms_worst_frame": worst,
            "neural_points_before": int(n_before), "neural_points_after": int(npts.count()),
            "points_appended_per_timed_frame": round((npts.count() - n_before - grown_warm) / max(steps, 1), 1),
            "local_points": int(npts.local_count()),
            "note": "same frame definition as `value`; the sensor moves, every frame has its own scan, the map grows and the "
                    "local map / brick cache follow the sensor"}


def gpu_parity_sample(npts, dec, trk, gp, xyz, k, n=4096, rgb=None, mp=None, cdec=None, n_train=2048):
    """Device side of bench.py's parity line, on the map as it stands after the timed frames: (i) the timed tracker kernels
    -- brick kNN as the tracker launches it, then the GN tile kernel with per-point outputs (+ the colour query of the
    photometric term) -- on `n` points of the timed scan; (ii) one training step of the timed shape -- pin_train_step
    (+ the colour step) as Mapper.mapping launches it -- on the first `n_train` samples of the pool, into scratch
    gradient / side-effect buffers.  Returns host copies of the outputs and of the map state the oracle needs."""
    import dataclasses
    from pin_slam_amd import ops
    cfg = npts.config
    q = xyz[:n].contiguous()
    nbr, nn, _ = npts.knn(q, True)
    fs = npts.field_state(dec, query_locally=True)
    ct, _keep = trk._color_term(rgb[:n]) if rgb is not None else (None, None)
    _, sdf, grad = ops.gn_accumulate(fs, gp, q, nbr, nn, want_points=True, color=ct)
    torch.cuda.synchronize()
    raw = nbr.cpu().numpy()[..., 3].view(np.int32)
    idx = np.where(raw >= 0, raw & ~0x40000000, raw)
    td = npts._travel()
    g = dict(q=q.cpu().numpy(), idx=idx, nn=nn.cpu().numpy(), sdf=sdf.cpu().numpy(), grad=grad.cpu().numpy(),
             table=npts.buffer_pt_index.cpu().numpy().astype(np.int64), pos=npts.neural_points.cpu().numpy(),
             ts_create=npts.point_ts_create.cpu().numpy(), travel=None if td is None else td.cpu().numpy(),
             cur_ts=int(npts.cur_ts), diff=float(npts.diff_travel_dist_local), g2l=npts.global2local.cpu().numpy(),
             lfeat=npts.local_geo_features.data.cpu().numpy(), lpos=npts.local_neural_points.cpu().numpy(),
             dec=dec.flat_params().cpu().numpy(), dx=npts.neighbor_dx.cpu().numpy(), mv=float(npts.max_valid_dist2),
             res=float(npts.resolution), valid_nn_k=int(gp.valid_nn_k), weighted_first=bool(cfg.weighted_first))
    fc = None
    if cdec is not None:
        fc = npts.field_state(cdec, query_locally=True, color=True)
        col, _, _ = ops.color_query(fc, q, nbr, nn, want_grad=False)
        g.update(color=col.cpu().numpy(), lcfeat=npts.local_color_features.data.cpu().numpy(), cdec=cdec.flat_params().cpu().numpy())
    if mp is not None and mp.pool_sample_count >= n_train:
        b = mp._pool().bufs[0]
        coord, label = b["global_coord"][:n_train].contiguous(), b["sdf_label"][:n_train].contiguous()
        wt, ts = b["weight"][:n_train].contiguous(), b["ts"][:n_train].contiguous()
        st = npts.search_state()
        cert, tsu = fs.certainty.clone(), npts.local_point_ts_update.clone()
        fst = dataclasses.replace(fs, certainty=cert, dec_image=None)
        gfeat, gdec = torch.zeros_like(fs.feats), torch.zeros_like(fs.dec)
        dec_n = int(cfg.gradient_decimation)
        eps = np.float32(cfg.voxel_size_m * cfg.num_grad_step_ratio)
        buf = ops.TrainBuffers(n_train, dec_n, k, fs.hidden, fs.levels, weighted_first=fs.weighted_first)
        bricks = npts._use_bricks()
        tf = bool(npts.temporal_local_map_on and npts.travel_dist is not None)
        if bricks is not None and (bricks.mode[:2] != (tf, True) or npts.neighbor_K != bricks.cand_dx.shape[0]):
            bricks = None
        loss = ops.train_step(st, fst, buf, coord, label, wt, ts, cert, tsu, gfeat, gdec, sigma=mp.sdf_scale, weight_e=cfg.weight_e,
                              eik_eps=eps, loss_weight_on=cfg.loss_weight_on, bricks=bricks)
        g["train"] = dict(coord=coord.cpu().numpy(), label=label.cpu().numpy(), weight=wt.cpu().numpy(), gfeat=gfeat.cpu().numpy(),
                          gdec=gdec.cpu().numpy(), loss=loss.cpu().numpy().copy(), n_eik=buf.n_eik, dec_n=dec_n, eps=float(eps),
                          weight_e=float(cfg.weight_e), loss_weight_on=bool(cfg.loss_weight_on), sigma=float(mp.sdf_scale))
        if fc is not None and cfg.weight_i > 0:
            clab = b["color"][:n_train, :3].contiguous()
            fct = dataclasses.replace(fc, certainty=cert, dec_image=None)
            gc, gcd = torch.zeros_like(fc.feats), torch.zeros_like(fc.dec)
            ops.train_color_step(fct, buf, label, clab, wt, gc, gcd, surface_range=cfg.surface_sample_range_m, weight_i=cfg.weight_i,
                                 loss_weight_on=cfg.loss_weight_on)
            g["train"].update(color_label=clab.cpu().numpy(), gcfeat=gc.cpu().numpy(), gcdec=gcd.cpu().numpy(),
                              surface_range=float(cfg.surface_sample_range_m), weight_i=float(cfg.weight_i))
        torch.cuda.synchronize()
    return g


def c4_single_gpu(args, cfg, mp, nn_mean=None, Kc=81, k=8):
    """Config C4 on one GPU: Mapper.mapping on a global batch of --global-bs samples (the quantity
    `bench.py --gpus N` reports for N > 1), timed with the host clock around synchronised calls."""
    bs0 = cfg.bs
    cfg.bs = args.global_bs
    try:
        mp.mapping(2)  # builds the trainer / workspace for this batch size
        torch.cuda.synchronize()
        reps = 2
        t0 = time.perf_counter()
        for _ in range(reps):
            mp.mapping(args.c4_iters)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        group = int(getattr(getattr(mp._trainer, "buf", None), "group", 1) or 1)
        want = getattr(mp, "reuse_pool_records", None)
        reused = bool(want) if want is not None else (args.c4_iters * args.global_bs >= getattr(mp, "reuse_pool_records_ratio", 2.0) * mp.pool_sample_count)
    finally:
        cfg.bs = bs0
        mp._trainer = None  # drop the 2^20-sample workspace
    it = reps * args.c4_iters
    out = {"global_batch": args.global_bs, "iterations_timed": it, "ms_per_iteration": round(1e3 * dt / it, 4),
           "mapper_samples_per_sec": round(args.global_bs * it / dt, 1), "optimizer": "lazy exact Adam (single GPU)"}
    out["pool_records_reused"] = reused
    out["roofline_train"] = roofline_train(args, cfg, out["ms_per_iteration"], nn_mean, Kc, k, group,
                                           reuse_pool_n=int(mp.pool_sample_count) if reused else None)
    return out


def roofline_train(args, cfg, ms_iter, nn_mean, Kc, k, group, reuse_pool_n=None):
    """The mapper iteration at the C4 batch against the HBM roofline: SURVEY 8(d)'s algorithmic bytes per sample (without the
    Adam term: the lazy optimiser's traffic is in the counters, not in the formula) x the batch, over the measured iteration;
    `traffic` = HBM-side bytes of the iteration's kernels from the PMC passes of this command (scripts/pmc_bench.sh c4 ->
    profiles/r03_pmc_c4.json; FETCH_SIZE of the wide streaming reads of the weight-gradient kernel corrected x2 as
    MI355X_MICROARCH.md prescribes for gfx950, the random-row kernels at face value)."""
    rho = nn_mean / Kc if nn_mean else 0.56  # (the C3 map's measured share of occupied candidate cells when not handed in)
    dec_n = max(1, int(cfg.gradient_decimation))
    bytes_s = (1.0 + 6.0 / dec_n) * (12 + 4 * Kc + 16 * rho * Kc + 36 * k + 64 * k) + 12
    alg = bytes_s * args.global_bs
    n_eik = (args.global_bs + dec_n - 1) // dec_n
    if reuse_pool_n:  # Mapper._pool_records: the samples' records are copied, the probes and (once per call) the pool searched
        knn_queries = 6 * n_eik + reuse_pool_n / max(1, args.c4_iters)
        knn_note = f"6 x {n_eik} probes per iteration + one search over the {reuse_pool_n} pool samples per call of {args.c4_iters} iterations"
    else:
        knn_queries = args.global_bs + 6 * n_eik
        knn_note = "every sample and probe, every iteration"
    r = {"kernel": "train_fused_kernel + train_dw_recompute_kernel + knn_brick_kernel + lazy Adam (one Mapper.mapping iteration, 2^20 samples)",
         "bound": "hbm", "achieved": round(alg / (ms_iter * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": round(alg / (ms_iter * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_sample": round(bytes_s, 1),
         "traffic": None}
    path = next((q for q in (os.path.join(ROOT, "profiles", f"{rnd}_pmc_c4.json") for rnd in ("r06", "r05", "r04", "r03")) if os.path.exists(q)), "")
    if path:
        try:
            ks = json.load(open(path))["kernels"]
            per = {"train_fused": ks["train_fused"]["hbm_bytes_per_launch"],
                   "train_dw": ks["train_dw_recompute" if "train_dw_recompute" in ks else "train_dw_stream"]["hbm_bytes_per_launch_if_streaming_x2"],
                   # the search kernel's bytes per QUERY (8 lanes per query) x the queries an iteration pays for
                   "knn_brick": int(ks["knn_brick"]["hbm_bytes_per_launch"] / (int(ks["knn_brick"]["launch_shape_grid"]) / 8.0) * knn_queries),
                   # (from three records per table row on the row-marking launch is skipped: no "mark_rows" class then)
                   "lazy_adam": ks.get("mark_rows", {}).get("hbm_bytes_per_launch", 0) + ks["adam_lazy_prepare_rows"]["hbm_bytes_per_launch"]}
            if reuse_pool_n:  # the copy of the samples' records (pin_gather_records_drawn) has no counter pass of its own: read + write
                per["gather_records_computed"] = int(2 * args.global_bs * k * 16)
            r["traffic"] = int(sum(per.values()))
            r["traffic_per_kernel"] = per
            r["traffic_source"] = "profiles/" + os.path.basename(path)
            r["traffic_over_algorithmic"] = round(r["traffic"] / alg, 2)
            r["kernel_us"] = {kk: ks[kk]["duration_us"] for kk in ("train_fused", "train_dw_recompute", "train_dw_stream", "knn_brick",
                                                                  "mark_rows", "adam_lazy_prepare_rows") if kk in ks}
            r["note"] = ("the operand stream of the weight gradient (deltas + decoder input, 1.15 KB per query, written by the tile kernel and "
                         "read back by the next launch; r03a streamed the layers' inputs as well, 2.2 KB) and the read-modify-write of the "
                         "gradient rows are the distance between traffic and algorithmic bytes; the search kernel is counted per query "
                         f"({knn_note})")
        except Exception:
            pass
    return r


def sync_replicas(npts, dec, world):
    """Set-up only: the single-GPU legs above trained every rank's replica with its own order of gradient atomics; make the
    replicas bit-identical again before the data-parallel run (rank 0's copy everywhere)."""
    if world <= 1:
        return
    import torch.distributed as dist
    host = dist.get_backend() == "gloo"  # (--dp-transport host: the ranks share one GPU)
    for t in (npts._g["geo"], npts._l["geo"], npts._g["cert"], npts._l["cert"], npts._g["ts_update"], npts._l["ts_update"],
              dec.flat_params()):
        if host:
            h = t.cpu()
            dist.broadcast(h, src=0)
            t.copy_(h)
        else:
            dist.broadcast(t, src=0)
    torch.cuda.synchronize()


# the exchange model of c4_per_rank_emulated (a ONE-GPU box cannot measure xGMI): ring / direct all-reduce of S bytes over W
# ranks = latency + 2 (W-1)/W * S / busbw; busbw for MB-sized messages taken well under the link peak (7 x 153 GB/s)
AR_LATENCY_US, AR_BUSBW_GBS = 30.0, 150.0


def allreduce_model_us(nbytes, world):
    return AR_LATENCY_US + 2.0 * (world - 1) / world * nbytes / (AR_BUSBW_GBS * 1e3)


def allgather_model_us(total_bytes, world):
    return AR_LATENCY_US + (world - 1) / world * total_bytes / (AR_BUSBW_GBS * 1e3)


def c4_per_rank_emulated(args, cfg, mp, worlds, single):
    """N = 1 box: single ranks of W-rank jobs, ALONE on this GPU (collective.NullComm: the identity exchange).  A rank of
    the spatially sharded mapper does its own share only -- its samples, the rows it owns, the whole halo -- so its time
    here is its compute time in the W-rank job; the all-reduces (halo per iteration, side effects + owner merge per call)
    are added from allreduce_model_us(), labelled as a model."""
    from pin_slam_amd import collective
    bs0 = cfg.bs
    cfg.bs = args.global_bs
    out = {}
    try:
        for W in worlds:
            ranks = list(range(W)) if W <= 4 else sorted({0, 1, W // 2, W - 1})
            per = []
            for r in ranks:
                mp.dp_rank, mp.dp_world, mp.dp_comm, mp.dp_mode = r, W, collective.NullComm(r, W), args.dp_mode
                mp._trainer = None
                mp.mapping(2)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(2):
                    mp.mapping(args.c4_iters)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                st = dict(getattr(mp, "dp_stats", None) or {})
                per.append(dict(rank=r, ms_per_iteration=round(1e3 * dt / (2 * args.c4_iters), 4), samples_max=st.get("samples_max"),
                                halo_fraction=round(st.get("halo_fraction", 0.0), 4), exchange_bytes=st.get("exchange_bytes"),
                                halo_rows=st.get("halo_rows"), merge=st.get("merge"), merge_bytes_per_call=st.get("merge_bytes_per_call")))
            slow = max(p["ms_per_iteration"] for p in per)
            xb = per[0]["exchange_bytes"] or (4 * int(mp._get_trainer().grad.numel()))
            rows = int(mp.neural_points.local_count())
            # per call.  spatial, merge "gather": one all-gather of the owned rows (features + certainty + ts), the halo's side
            # effects as two small all-reduces; merge "reduce" / dense: certainty (fp32) + ts (int32) all-reduces over every
            # row, and (spatial) the all-reduce of the table
            if per[0].get("merge") == "gather":
                per_call_us = allgather_model_us(40 * W * max(1, (per[0]["merge_bytes_per_call"] - 12 * per[0]["halo_rows"]) // (40 * W)), W) \
                    + 2 * allreduce_model_us(4 * per[0]["halo_rows"], W)
            else:
                per_call_us = 2 * allreduce_model_us(4 * rows, W) + (allreduce_model_us(32 * rows, W) if args.dp_mode == "spatial" else 0.0)
            # r04: the all-reduce runs on a side stream in two messages (engine.MapTrainer.step_batch) -- the halo rows beside the
            # weight-gradient launch of the same iteration and the lazy-Adam launch of the next one, the decoder's 53 KB beside
            # the lazy-Adam launch.  The measured rank time already contains the event waits of that structure (the exchange
            # itself is the identity here); what the model adds is the part of each message that outlasts the work it hides
            # behind: shares of the rank's iteration from profiles/r04_dp_rank_8_0_kernel_stats.csv (weight gradient 0.23, lazy
            # Adam 0.09 of the iteration).
            nd_bytes = 4 * int(mp._get_trainer().dp.nd) if getattr(mp._get_trainer(), "dp", None) is not None else 0
            overlapped = os.environ.get("PIN_DP_OVERLAP", "1") != "0" and args.dp_mode == "spatial"
            if overlapped:
                halo_us, dec_us = allreduce_model_us(max(xb - nd_bytes, 0), W), allreduce_model_us(nd_bytes, W)
                exposed_us = max(0.0, halo_us - 0.32 * slow * 1e3) + max(0.0, dec_us - 0.09 * slow * 1e3)
            else:
                halo_us, dec_us = allreduce_model_us(xb, W), 0.0
                exposed_us = halo_us
            comm_ms = (exposed_us + per_call_us / args.c4_iters) * 1e-3
            proj = args.global_bs / ((slow + comm_ms) * 1e-3)
            out[str(W)] = dict(ranks_measured=per, slowest_rank_ms_per_iteration=slow, exchange_bytes_per_iteration=xb,
                               exchange_overlapped=overlapped, modelled_allreduce_us={"halo_rows": round(halo_us, 1), "decoder": round(dec_us, 1),
                                                                                      "exposed": round(exposed_us, 1)},
                               modelled_exchange_ms_per_iteration=round(comm_ms, 4),
                               projected_samples_per_sec=round(proj, 1),
                               projected_speedup_vs_single_gpu=None if not single else round(proj / single["mapper_samples_per_sec"], 2))
    finally:
        cfg.bs = bs0
        mp.dp_rank, mp.dp_world, mp.dp_comm = 0, 1, None
        mp._trainer = None
    out["model"] = (f"PROJECTED, not measured: slowest measured rank + the part of the modelled all-reduces ({AR_LATENCY_US:.0f} us + 2(W-1)/W * bytes / "
                    f"{AR_BUSBW_GBS:.0f} GB/s; halo rows and decoder as two messages on a side stream) that outlasts the launches they run beside "
                    f"(weight gradient + next lazy-Adam launch: 0.32 of the rank's iteration; decoder message: the lazy-Adam launch, 0.09) + per call of "
                    f"{args.c4_iters} iterations the owner merge (all-gather {AR_LATENCY_US:.0f} us + (W-1)/W * bytes / {AR_BUSBW_GBS:.0f} GB/s) and the "
                    f"halo's side effects; ranks run alone on one GPU with the identity exchange (dp_mode {args.dp_mode})")
    return out


def bench_single_gpu_mapper(args, cfg, mp, npts, wl, P):
    """`--gpus 1 --metric mapper`: Mapper.mapping on the global batch on ONE GPU, timed exactly like the N > 1 step of
    bench_dp_mapper (warm-up calls, then --steps calls of --map-iters iterations between synchronisations)."""
    H, L = wl["hidden"], wl["levels"]
    gbs, iters = args.global_bs, args.map_iters
    cfg.bs = gbs
    torch.manual_seed(4242)
    for _ in range(max(1, args.warmup)):
        mp.mapping(iters)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        mp.mapping(iters)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    value = args.steps * iters * gbs / elapsed
    return {"metric": "mapper_samples_per_sec", "value": round(value, 1), "unit": "samples/s", "n_gpus": 1, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"c4: Mapper.mapping, global batch {gbs} (+{(gbs + 9) // 10}x6 Eikonal probes) on 1 GPU, {iters} "
                                   f"iterations per step, {wl['desc']}",
                       "neural_points": P, "local_points": int(npts.local_count()), "decoder": f"{L}x{H}", "global_batch": gbs,
                       "per_rank_batch": gbs, "dp_mode": None,
                       "parallelism": "one GPU: lazy exact Adam, pool records reused across the call; the N = 1 point of the "
                                      "`bench.py --gpus N` strong-scaling curve (same metric, same step)"},
            "ms_per_iteration": round(1e3 * elapsed / (args.steps * iters), 4), "rccl_ranks": 0, "scaling_efficiency": 1.0,
            "allreduce": None}


def bench_dp_mapper(args, cfg, mp, npts, wl, rank, world, barrier, max_over_ranks, P, single):
    """N > 1: the data-parallel mapper (config C4).  Step = one Mapper.mapping call of --map-iters iterations on the
    global batch.  spatial: every rank trains the samples in its k-d box, lazy Adam on its rows, one RCCL all-reduce of
    [decoder | halo rows] per iteration; dense: contiguous shards, all-reduce of the whole gradient, replicated dense Adam."""
    H, L = wl["hidden"], wl["levels"]
    gbs, iters = args.global_bs, args.map_iters
    sync_replicas(npts, mp.sdf_mlp, world)
    # every rank draws the batches of a call from its own generator: put them in the same state whatever each rank's history was
    # (the spatial mapper compares a checksum of the first drawn batch across the ranks and refuses to run on diverged replicas)
    torch.manual_seed(4242)
    ar_pairs, cur = [], {}
    n_ev = 2 * iters
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_ev)]

    def on_ar(start):
        if start:
            if len(ar_pairs) < n_ev:
                cur["p"] = evs[len(ar_pairs)]
                cur["p"][0].record()
            else:
                cur["p"] = None
        elif cur.get("p") is not None:
            cur["p"][1].record()
            ar_pairs.append(cur["p"])

    for _ in range(max(1, args.warmup)):
        mp.mapping(iters)
    barrier()
    mp._get_trainer().on_allreduce = on_ar
    t0 = time.perf_counter()
    for i in range(args.steps):
        mp.mapping(iters)
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    t = mp._get_trainer()
    t.on_allreduce = None
    spatial = t.dp is not None
    st = dict(getattr(mp, "dp_stats", None) or {})
    ar_ms = float(np.mean([a.elapsed_time(b) for a, b in ar_pairs])) if ar_pairs else float("nan")
    ar_bytes = int(st["exchange_bytes"]) if spatial else 4 * int(t.grad.numel())
    overlapped = bool(spatial and getattr(t, "overlap_exchange", False))
    if overlapped:  # the events bracket the halo-row message (side stream); the decoder's follows as a second, small message
        ar_bytes -= 4 * int(t.dp.nd)
    ms_it = 1e3 * elapsed / (args.steps * iters)
    value = args.steps * iters * gbs / elapsed
    # the slowest rank's share of the samples (load balance of the boxes)
    smax = max_over_ranks(float(st.get("samples_max", gbs // world)))
    ok = ar_ms == ar_ms and ar_ms > 0
    busbw = 2 * (world - 1) / world * ar_bytes / (ar_ms * 1e-3) / 1e9 if ok else None
    return {
        "metric": "mapper_samples_per_sec", "value": round(value, 1), "unit": "samples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"c4: Mapper.mapping, global batch {gbs} (+{(gbs + 9) // 10}x6 Eikonal probes) over "
                               f"{world} GPUs, {iters} iterations per step, {wl['desc']}",
                   "neural_points": P, "local_points": int(npts.local_count()), "decoder": f"{L}x{H}",
                   "global_batch": gbs, "per_rank_batch": gbs // world, "dp_mode": args.dp_mode,
                   "parallelism": (f"mapper dp{world}, spatial shards: each rank trains the samples of every drawn batch inside its k-d box "
                                   f"of the voxel grid, lazy exact Adam on the rows it owns; per iteration ONE RCCL all-reduce "
                                   f"(pin_allreduce_f32, on a side stream: halo rows beside the weight gradient, decoder beside the next lazy-Adam launch) of [decoder | halo-row] gradients + the same "
                                   f"dense Adam step on the halo rows everywhere; owned rows, certainty and ts published once per call")
                                  if spatial else
                                  (f"mapper dp{world}, dense: contiguous batch shards, map + decoder replicated, one RCCL all-reduce "
                                   f"(pin_allreduce_grads) of [decoder | feature] gradients per iteration, replicated dense Adam; "
                                   f"certainty / ts merged once per call"),
                   "n1_reference": "single_gpu_same_box below; the N = 1 bench line reports it as c4_single_gpu"},
        "ms_per_iteration": round(ms_it, 4),
        # what ran the exchange, at top level: "rccl" = pin_slam_amd.collective.RcclComm (RCCL through the C ABI), nothing else
        "transport": _transport_tag(mp.dp_comm), "rccl_ranks": world if _transport_tag(mp.dp_comm) == "rccl" else 0,
        "busbw_GBs": round(busbw, 1) if ok else None,
        "single_gpu_same_box": single,
        "speedup_vs_single_gpu": None if not single else round(value / single["mapper_samples_per_sec"], 3),
        # strong scaling: N ranks against N x the one-GPU mapper measured by every rank of THIS run on its own GPU
        "scaling_efficiency": None if not single else round(value / (world * single["mapper_samples_per_sec"]), 4),
        "shards": None if not spatial else {"halo_rows": st.get("halo_rows"), "rows": st.get("rows"),
                                            "halo_fraction": round(st.get("halo_fraction", 0.0), 4),
                                            "largest_share_of_batch": round(smax / gbs, 4), "ideal_share": round(1.0 / world, 4)},
        "allreduce": {"transport": getattr(mp.dp_comm, "kind", None), "bytes_per_iteration": ar_bytes,
                      "overlapped": overlapped,
                      "note": ("halo-row message on a side stream, beside the weight gradient of the same iteration and the lazy-Adam launch "
                               "of the next; the decoder gradients follow as a second message; share_of_iteration is the message's "
                               "duration over the iteration, NOT time added to it") if overlapped else
                              "in place on the compute stream",
                      "avg_ms": round(ar_ms, 4) if ok else None, "launches_timed": len(ar_pairs),
                      "algbw_GBs": round(ar_bytes / (ar_ms * 1e-3) / 1e9, 1) if ok else None,
                      "busbw_GBs": round(busbw, 1) if ok else None,
                      "share_of_iteration": round(ar_ms / ms_it, 3) if ok else None},
        "roofline": {"kernel": "ncclAllReduce (xGMI) of the per-iteration exchange buffer", "bound": "xgmi-link",
                     "achieved": round(busbw, 1) if ok else None, "peak": 7 * 153.0, "unit": "GB/s",
                     "frac": round(busbw / (7 * 153.0), 4) if ok else None, "traffic": ar_bytes,
                     "note": "bus bandwidth of the per-iteration all-reduce per GPU against 7 xGMI links x 153 GB/s; with "
                             "spatial shards the message is a few MB and latency-bound by design -- the lever is its SIZE "
                             "(bytes_per_iteration against the 71 MB of the dense exchange), see share_of_iteration"},
    }


def _transport_tag(comm) -> str:
    from pin_slam_amd import collective
    if isinstance(comm, collective.RcclComm):
        return "rccl"
    if isinstance(comm, collective.TorchComm):
        return "torch.distributed"
    if isinstance(comm, collective.HostStagedComm):
        return "host-staged (test mode)"
    return "none"


def _ref_cpu_live(args):
    """cpu_baseline timed by THIS command: when the reference pack travels with the snapshot (oracle/_ref/, written by
    `scripts/e2e_pin_slam.py pack`; git-ignored, never part of the repository), scripts/ref_cpu_baseline.py runs the
    unmodified reference classes on torch CPU on this host -- after the timed region, the GPU idle -- with as many threads as
    the container's CPU quota allows (more are throttled: profiles/r04_ref_cpu_baseline.json holds the sweep).  None when the
    pack is absent or the run fails: the committed record stays the baseline."""
    import subprocess
    pack = os.path.join(ROOT, "oracle", "_ref", "pin_slam_reference.tar.gz")
    if not os.path.exists(pack) or os.environ.get("PIN_BENCH_REF_LIVE", "1") == "0":
        return None
    from pin_slam_amd.dropin import cpu_quota
    # threads: the container's CPU quota, but never more than 16 -- the reference's own thread sweep on these hosts
    # (profiles/r04_ref_cpu_baseline.json: 16 / 32 / 64 / 128) has 16 as the fastest setting and torch's default (every
    # logical CPU of an unconstrained host) as the slowest: more threads would only inflate the GPU / CPU ratio
    threads = max(1, min(16, int(cpu_quota())))
    dst = os.path.join(ROOT, "gpurun_out")
    os.makedirs(dst, exist_ok=True)
    rec = os.path.join(dst, "ref_cpu_baseline_live.json")
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "ref_cpu_baseline.py"), "--reps", "5", "--tracking-reps", "1", "--threads",
           str(threads), "--reg-iters", str(args.reg_iters), "--map-iters", str(args.map_iters), "--out", rec,
           "--host-label", "host of this bench run"]
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    try:
        subprocess.run(cmd, check=True, timeout=300, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=env)
        r = json.load(open(rec))
    except Exception:
        return None
    return {"value": r["frames_per_sec_bench_definition"], "unit": "frames/s", "cores": r["torch_threads"],
            "kind": "reference-torch-cpu", "live": True, "host": r["host"],
            "sample": (f"unmodified reference classes on torch CPU timed by this command after the GPU legs: median of {r['reps']} x "
                       f"Tracker.registration_step over the {r['scan_points']}-point scan ({r['registration_step_ms']} ms) and of "
                       f"{r['reps']} x Mapper.mapping({r['mapping_iterations']}) ({r['mapping_ms']} ms), 1 warm-up each; frame = "
                       f"{args.reg_iters} registration steps + one mapping call (preprocess / map prep not included, which favours "
                       f"the CPU); {r['torch_threads']} torch threads = min(16, this container's CPU quota); one Tracker.tracking call "
                       f"(its own convergence test, no warm-up) timed beside them: tracking_ms"),
            "value_range": r.get("frames_per_sec_range"),
            "spread_ms": {"registration_step": [min(r["registration_step_ms_all"]), r["registration_step_ms"], max(r["registration_step_ms_all"])],
                          "mapping": [min(r["mapping_ms_all"]), r["mapping_ms"], max(r["mapping_ms_all"])],
                          "order": "min, median, max of the repeats"} if "registration_step_ms_all" in r else None,
            "tracking_ms": r.get("tracking_ms"),
            "registration_queries_per_sec": r["registration_queries_per_sec"],
            "mapper_samples_per_sec": r["mapper_samples_per_sec"], "torch": r.get("torch")}


def _host_now():
    """CPU model / logical CPU count of the host this command runs on (the cpu_baseline record names its own)."""
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    cpus = os.cpu_count() or 0
    return {"model": model, "cpus": cpus, "text": f"{cpus} logical CPUs, {model or 'unknown CPU'}"}


def cpu_baseline(m, cfg, scan, raw, raw_ts, pool_c, pool_l, feats, dec, H, L, k, sdf_scale, args):
    """The numpy oracle (a port of the reference's torch-CPU chain) timed on a bounded sample of
    the same workload on this host; odometry and mapping are extrapolated linearly in the query
    count, the preprocess / map-prep stages are timed at full size (they are cheap)."""
    from oracle import pin_oracle as O
    n_s, bs_s = 20000, 4096
    # the oracle works on the synthetic map arrays directly (same voxels / hash as the device map)
    params = O.unpack_decoder(dec, 11, H, L)
    dx, mv = O.search_neighborhood(cfg.num_nei_cells, cfg.search_alpha, m.resolution)
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
                     (11, H, L), sdf_scale, k, dec=10, eps=cfg.voxel_size_m * cfg.num_grad_step_ratio, weight_e=cfg.weight_e,
                     dtype=np.float32)

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
        coord, label, _, w = O.sample_rays(pc[:, :3], None, g.standard_normal(cfg.surface_sample_n * n, dtype=np.float32),
                                           g.random(cfg.free_front_n * n, dtype=np.float32),
                                           g.random(cfg.free_behind_n * n, dtype=np.float32),
                                           surface_range=cfg.surface_sample_range_m, surface_n=cfg.surface_sample_n,
                                           front_n=cfg.free_front_n, behind_n=cfg.free_behind_n,
                                           free_begin_ratio=cfg.free_sample_begin_ratio, free_end_dist=cfg.free_sample_end_dist_m,
                                           max_range=cfg.max_range)
        glob = np.concatenate([pool_c, coord])
        O.pool_filter_mask(glob, np.zeros(3), cfg.window_radius)
        st = dict(table=table64.copy(), positions=m.positions, ts_create=np.zeros(len(m.positions), np.int32),
                  ts_update=np.zeros(len(m.positions), np.int32))
        O.map_update(st, coord[np.abs(label) < cfg.surface_sample_range_m * cfg.map_surface_ratio], 1, m.resolution, temporal=False)
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
    base = {"value": round(1.0 / frame_s, 5), "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"numpy oracle: {reps} registration steps on {n_s} scan points ({t_reg*1e3:.0f} ms each) and "
                      f"{reps2} mapping iterations of batch {bs_s} ({t_tr*1e3:.0f} ms each), extrapolated linearly to "
                      f"{args.reg_iters}x{args.scan} + {args.map_iters}x{args.bs}; preprocess + map-prep stages once at "
                      f"full size ({t_prep:.1f} s); one host thread (host has {os.cpu_count()} cores)",
            "registration_queries_per_sec": round(n_s / t_reg, 1), "mapper_samples_per_sec": round(bs_s / t_tr, 1)}
    return base


def parity_vs_oracle(O, g, H, L, k, sdf_scale):
    """The oracle's answer to the queries of gpu_parity_sample() on the same map state (the checker; never timed)."""
    wf = g["weighted_first"]

    def search(p):
        return O.radius_search(p, g["table"], g["pos"], g["res"], g["dx"], g["mv"], ts_create=g["ts_create"],
                               travel_dist=g["travel"], cur_ts=g["cur_ts"], diff_travel_dist_local=g["diff"])

    s = search(g["q"])
    qf = O.query_feature(g["q"], s, g["lfeat"], g["lpos"], None, k, global2local=g["g2l"])
    params = O.unpack_decoder(g["dec"], 11, H, L)
    rs, rg, _, rnn, _ = O.query_sdf(g["q"], s, g["lfeat"], g["lpos"], params, sdf_scale, k, weighted_first=wf, global2local=g["g2l"])
    has = rnn >= g["valid_nn_k"]
    scale = float(np.abs(rs[has]).max()) if has.any() else 1.0
    gscale = np.abs(rg).max(1, keepdims=True) + 1e-6
    out = {"n_checked": int(len(g["q"])), "n_with_neighbours": int(has.sum()),
           "idx_mismatch": int((g["idx"] != qf["knn_idx"].astype(np.int32)).sum()),
           "nn_count_mismatch": int((g["nn"] != rnn).sum()),
           "sdf_max_abs": float(np.abs(g["sdf"] - rs)[has].max()) if has.any() else None,
           "sdf_max_rel": float(np.abs(g["sdf"] - rs)[has].max() / scale) if has.any() else None,
           "grad_max_rel": float((np.abs(g["grad"] - rg) / gscale)[has].max()) if has.any() else None,
           "definition": "brick kNN + GN tile kernel (the timed kernels) vs the numpy oracle (float64 decoder) on the first "
                         "n_checked points of the timed scan and the map after the timed frames; sdf_max_rel = max |sdf - ref| "
                         "/ max |ref|, grad_max_rel = max over points of |grad - ref| / max-component of ref; train_*: one "
                         "pin_train_step (+ colour step) of the timed shape on the first n_train pool samples, max |g - ref| / "
                         "max |ref| over the feature rows / decoder parameters, relative error of the two loss terms; target 1e-4"}
    if "color" in g:  # Decoder.regress_color of the photometric term (tracker.py:342-350)
        cpar = O.unpack_decoder(g["cdec"], 11, H, L, 3)
        rc, _, _ = O.query_color(g["q"], s, g["lcfeat"], g["lpos"], cpar, k, weighted_first=wf, global2local=g["g2l"])
        out["color_max_abs"] = float(np.abs(g["color"] - rc)[rnn > 0].max())
    t = g.get("train")
    if t is not None:
        def searcher(table_feats):
            return lambda p: O.query_feature(p, search(p), table_feats, g["lpos"], None, k, global2local=g["g2l"], weighted_first=False)
        r = O.train_step(t["coord"], t["label"], t["weight"], searcher(g["lfeat"]), g["lfeat"], g["lpos"], g["dec"], (11, H, L),
                         t["sigma"], k, weighted_first=wf, dec=t["dec_n"], eps=t["eps"], weight_e=t["weight_e"],
                         loss_weight_on=t["loss_weight_on"])
        n_tr = len(t["label"])
        out.update({"n_train": n_tr,
                    "train_feat_grad_max_rel": float(np.abs(t["gfeat"] - r["feat_grad"]).max() / np.abs(r["feat_grad"]).max()),
                    "train_dec_grad_max_rel": float(np.abs(t["gdec"] - r["dec_grad"]).max() / np.abs(r["dec_grad"]).max()),
                    "train_rows_touched_mismatch": int(((np.abs(t["gfeat"]).max(1) > 0) != (np.abs(r["feat_grad"]).max(1) > 0)).sum()),
                    "train_bce_loss_rel": float(abs(t["loss"][0] / n_tr - r["sdf_loss"]) / abs(r["sdf_loss"])),
                    "train_eik_loss_rel": float(abs(t["loss"][1] / max(t["n_eik"], 1) - r["eik_loss"]) / abs(r["eik_loss"]))
                                          if t["n_eik"] and r["eik_loss"] else None})
        if "gcfeat" in t:
            rc = O.train_color_step(t["coord"], t["label"], t["color_label"], t["weight"], searcher(g["lcfeat"]), g["lcfeat"], g["cdec"],
                                    (11, H, L, 3), k, weighted_first=wf, surface_range=t["surface_range"], weight_i=t["weight_i"],
                                    loss_weight_on=t["loss_weight_on"])
            out.update({"train_color_feat_grad_max_rel": float(np.abs(t["gcfeat"] - rc["feat_grad"]).max() / np.abs(rc["feat_grad"]).max()),
                        "train_color_dec_grad_max_rel": float(np.abs(t["gcdec"] - rc["dec_grad"]).max() / np.abs(rc["dec_grad"]).max())})
    return out


if __name__ == "__main__":
    main()
