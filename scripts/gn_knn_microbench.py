"""kNN (brick cache) and GN tile kernel timed alone on the bench's C3 map for several query counts (HIP events over 30
launches each): separates the per-launch constant from the per-tile slope.  `PIN_LIB=<path>` loads another build of
libpinhip.so (A/B runs in one gpurun call); the usual PIN_* variant switches apply."""
import os, sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pin_slam_amd._lib as _L  # noqa: E402
_L.LIB_PATH = os.environ.get("PIN_LIB", _L.LIB_PATH)
from pin_slam_amd import ops, synth  # noqa: E402
from pin_slam_amd._lib import GnParams  # noqa: E402

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 16
# optional: one query count and a repeat count (profiling runs: one launch shape per kernel), e.g. `16 98756 50`
only_n = int(sys.argv[2]) if len(sys.argv) > 2 else None
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
H, L = (64, 4) if layers >= 16 else (32, 2)
m = synth.build_map(layers=layers)
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
fs = ops.FieldState(feats=dev(m.features), dec=dev(synth.init_decoder(H, L)), k=8, hidden=H, levels=L,
                    weighted_first=True, sdf_scale=0.055, certainty=torch.zeros(P, device="cuda"), pos=pos)
scan = dev(synth.make_scan(m))
scan = ops.spatial_sort(scan.contiguous(), 0.1)  # the order Tracker.tracking registers its points in (engine.GNTracker.track)
gp = GnParams(); gp.valid_nn_k = 8; gp.min_grad_norm = 0.5; gp.max_grad_norm = 2.0; gp.max_sdf_std = 0.25; gp.gm_dist = 0.3; gp.gm_grad = 0.1
bricks = ops.BrickCache(dx, 2).build(st, wait=True)
fs.stage_decoder()  # as the tracker does once per registration


def timeit(fn, n=reps):
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


print("lib", _L.LIB_PATH)
for n in ((only_n,) if only_n else (16 * 3072, 16 * 6144, 98756, 16 * 9216)):
    s = torch.cat([scan, scan[: n - scan.shape[0]]]).contiguous() if n > scan.shape[0] else scan[:n].contiguous()
    nbr, nn, cur = ops.knn_query(st, s, 8, pose=np.eye(4), bricks=bricks)
    t_k = timeit(lambda: ops.knn_query(st, s, 8, pose=np.eye(4), out=(nbr, nn, cur), bricks=bricks))
    t_g = timeit(lambda: ops.gn_accumulate(fs, gp, cur, nbr, nn))
    print(f"n={n:7d} tiles/simd={n / 16 / 1024:6.3f}  knn {t_k:6.1f} us   gn {t_g:6.1f} us   gn per tile/simd {t_g / (n / 16 / 1024):6.2f}")
