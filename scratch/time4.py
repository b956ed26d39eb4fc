import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pin_slam_amd import ops, synth
from pin_slam_amd._lib import GnParams
m = synth.build_map(layers=16)
P = len(m.positions)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
pos = dev(m.positions); pos4 = torch.empty((P, 4), dtype=torch.float32, device="cuda")
ops.pack_positions(pos, torch.zeros(P, dtype=torch.int32, device="cuda"), pos4)
dx, mv = ops.search_neighborhood(2, 0.5, 0.4)
g2l = torch.arange(P + 1, dtype=torch.int32, device="cuda"); g2l[-1] = -1
st = ops.SearchState(table=dev(m.table), pos4=pos4, cand_off=dev(ops.candidate_offsets(dx, m.buffer_size)), n_points=P,
                     resolution=0.4, max_valid_dist2=mv, travel_dist=torch.zeros(1, device="cuda"), cur_ts=0,
                     diff_travel_dist_local=400.0, global2local=g2l)
gp = GnParams(); gp.valid_nn_k = 8; gp.min_grad_norm = 0.5; gp.max_grad_norm = 2.0; gp.max_sdf_std = 0.25; gp.gm_dist = 0.3; gp.gm_grad = 0.1
def ev(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); [fn() for _ in range(n)]; b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
feats = dev(m.features); cert = torch.zeros(P, device="cuda")
N = 100_000
scan = dev(synth.make_scan(m, n=N))
key = torch.floor(scan / 0.4).long()
scan = scan[torch.argsort((key[:, 0] + 4096) + ((key[:, 1] + 4096) << 14) + ((key[:, 2] + 4096) << 28))].contiguous()
nbr, nn, cur = ops.knn_query(st, scan, 8, pose=np.eye(4))
sums = torch.empty((64, 32), dtype=torch.float64, device="cuda")
for (H, L) in ((64, 4), (64, 2), (64, 1)):
    fs = ops.FieldState(feats=feats, dec=dev(synth.init_decoder(H, L)), k=8, hidden=H, levels=L, weighted_first=True,
                        sdf_scale=0.055, certainty=cert, pos=pos)
    t_gn = ev(lambda: ops.gn_accumulate(fs, gp, cur, nbr, nn, sums=sums))
    print(f"GN={os.environ.get('PIN_GN','quad')} DBG={os.environ.get('PIN_GQ_DBG','0')} N={N} {L}x{H}: gn {t_gn:7.1f} us", flush=True)
