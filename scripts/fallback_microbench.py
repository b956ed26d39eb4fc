"""The Gauss-Newton accumulation on the configurations that do NOT reach a tile kernel (per-neighbour decoding with a decoder
of more than one layer: gn_accumulate_mfma_kernel, 64 queries per wave) timed beside the tile kernels on the bench's C3 map:
what a caller outside the shipped configurations pays.  usage (GPU box): python scripts/fallback_microbench.py"""
import os, sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pin_slam_amd import ops, synth  # noqa: E402
from pin_slam_amd._lib import GnParams  # noqa: E402

m = synth.build_map(layers=16)
P = len(m.positions)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
pos = dev(m.positions)
pos4 = torch.empty((P, 4), dtype=torch.float32, device="cuda")
ops.pack_positions(pos, torch.zeros(P, dtype=torch.int32, device="cuda"), pos4)
dx, mv = ops.search_neighborhood(2, 0.5, 0.4)
g2l = torch.arange(P + 1, dtype=torch.int32, device="cuda"); g2l[-1] = -1
st = ops.SearchState(table=dev(m.table), pos4=pos4, cand_off=dev(ops.candidate_offsets(dx, m.buffer_size)), n_points=P,
                     resolution=0.4, max_valid_dist2=mv, travel_dist=torch.zeros(1, device="cuda"), cur_ts=0,
                     diff_travel_dist_local=410.0, global2local=g2l)
scan = ops.spatial_sort(dev(synth.make_scan(m)).contiguous(), 0.1)
gp = GnParams(); gp.valid_nn_k = 6; gp.min_grad_norm = 0.5; gp.max_grad_norm = 2.0; gp.max_sdf_std = 0.25; gp.gm_dist = 0.3; gp.gm_grad = 0.1
bricks = ops.BrickCache(dx, 2).build(st, wait=True)
nbr, nn, cur = ops.knn_query(st, scan, 8, pose=np.eye(4), bricks=bricks)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for wf, H, L, what in ((True, 64, 4, "tile kernel (weighted-first)"), (False, 64, 1, "tile kernel (per-neighbour, one layer)"),
                       (False, 64, 2, "FALLBACK gn_accumulate_mfma_kernel (per-neighbour, two layers)"),
                       (False, 32, 3, "FALLBACK gn_accumulate_mfma_kernel (per-neighbour, three layers of 32)")):
    fs = ops.FieldState(feats=dev(m.features), dec=dev(synth.init_decoder(H, L)), k=8, hidden=H, levels=L, weighted_first=wf,
                        sdf_scale=0.055, certainty=torch.zeros(P, device="cuda"), pos=pos)
    fs.stage_decoder()
    t = timeit(lambda: ops.gn_accumulate(fs, gp, cur, nbr, nn))
    print(f"{scan.shape[0]} queries, k = 8, decoder {L}x{H}, weighted_first {wf}: {t:7.1f} us   {what}")
