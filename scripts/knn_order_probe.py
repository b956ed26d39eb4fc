"""kNN (brick cache) + GN tile kernel timing against the ORDER of the scan points (same points, same map): does a
per-frame spatial sort of the registration points pay?"""
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
fs = ops.FieldState(feats=dev(m.features), dec=dev(synth.init_decoder(64, 4)), k=8, hidden=64, levels=4,
                    weighted_first=True, sdf_scale=0.055, certainty=torch.zeros(P, device="cuda"), pos=pos)
scan0 = dev(synth.make_scan(m))
gp = GnParams(); gp.valid_nn_k = 8; gp.min_grad_norm = 0.5; gp.max_grad_norm = 2.0; gp.max_sdf_std = 0.25; gp.gm_dist = 0.3; gp.gm_grad = 0.1
bricks = ops.BrickCache(dx, 2).build(st, wait=True)
fs.stage_decoder()


def key_lin(p, res):
    k = torch.floor(p / res).long() + 4096
    return k[:, 0] + (k[:, 1] << 14) + (k[:, 2] << 28)


def spread(v):  # 10 bits -> every third bit
    v = (v | (v << 16)) & 0x030000FF
    v = (v | (v << 8)) & 0x0300F00F
    v = (v | (v << 4)) & 0x030C30C3
    return (v | (v << 2)) & 0x09249249


def key_morton(p, res):
    k = (torch.floor(p / res).long() + 512) & 1023
    return spread(k[:, 0]) | (spread(k[:, 1]) << 1) | (spread(k[:, 2]) << 2)


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


orders = {
    "as generated": torch.arange(scan0.shape[0], device="cuda"),
    "shuffled": torch.randperm(scan0.shape[0], device="cuda"),
    "voxel 0.08 m, x fastest (reference down-sampler order)": torch.argsort(key_lin(scan0, 0.08)),
    "voxel 0.4 m, x fastest": torch.argsort(key_lin(scan0, 0.4)),
    "brick 1.6 m x fastest, then cell": torch.argsort(key_lin(scan0, 1.6) * 64 + (key_lin(scan0, 0.4) & 63)),
    "morton 0.4 m": torch.argsort(key_morton(scan0, 0.4)),
    "morton 0.1 m": torch.argsort(key_morton(scan0, 0.1)),
}
for name, perm in orders.items():
    s = scan0[perm].contiguous()
    nbr, nn, cur = ops.knn_query(st, s, 8, pose=np.eye(4), bricks=bricks)
    t_k = timeit(lambda: ops.knn_query(st, s, 8, pose=np.eye(4), out=(nbr, nn, cur), bricks=bricks))
    t_g = timeit(lambda: ops.gn_accumulate(fs, gp, cur, nbr, nn))
    print(f"{name:55s} knn {t_k:6.1f} us   gn {t_g:6.1f} us")
