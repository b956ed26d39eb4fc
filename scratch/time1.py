import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pin_slam_amd import ops, synth
from pin_slam_amd._lib import GnParams
layers = int(sys.argv[1]) if len(sys.argv) > 1 else 16
H, L = (64, 4) if layers >= 16 else (32, 2)
t0 = time.time(); m = synth.build_map(layers=layers); print("map", m.positions.shape, time.time()-t0, flush=True)
P = len(m.positions)
dev = lambda a, dt=None: (torch.from_numpy(np.ascontiguousarray(a)).to(dt) if dt else torch.from_numpy(np.ascontiguousarray(a))).cuda()
pos = dev(m.positions); ts = torch.zeros(P, dtype=torch.int32, device="cuda")
pos4 = torch.empty((P, 4), dtype=torch.float32, device="cuda"); ops.pack_positions(pos, ts, pos4)
dx, mv = ops.search_neighborhood(2, 0.5, 0.4)
st = ops.SearchState(table=dev(m.table), pos4=pos4, cand_off=dev(ops.candidate_offsets(dx, m.buffer_size)), n_points=P,
                     resolution=0.4, max_valid_dist2=mv, travel_dist=torch.zeros(1, device="cuda"), cur_ts=0,
                     diff_travel_dist_local=400.0, global2local=torch.arange(P + 1, dtype=torch.int32, device="cuda"))
st.global2local[-1] = -1
fs = ops.FieldState(feats=dev(m.features), dec=dev(synth.init_decoder(H, L)), k=8, hidden=H, levels=L, weighted_first=True,
                    sdf_scale=0.055, certainty=torch.zeros(P, device="cuda"), pos=pos)
scan = dev(synth.make_scan(m))
gp = GnParams(); gp.valid_nn_k = 8; gp.min_grad_norm = 0.5; gp.max_grad_norm = 2.0; gp.max_sdf_std = 0.25; gp.gm_dist = 0.3; gp.gm_grad = 0.1
def ev(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); [fn() for _ in range(n)]; b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
out = None
nbr, nn, cur = ops.knn_query(st, scan, 8, pose=np.eye(4))
out = (nbr, nn, cur)
print("nn mean", nn.float().mean().item(), "rho", nn.float().mean().item()/81)
t_knn = ev(lambda: ops.knn_query(st, scan, 8, pose=np.eye(4), out=out))
sums = torch.empty((64, 32), dtype=torch.float64, device="cuda")
t_gn = ev(lambda: ops.gn_accumulate(fs, gp, cur, nbr, nn, sums=sums))
t_sdf = ev(lambda: ops.sdf_query(fs, cur, nbr, nn))
# sorted-by-voxel scan
key = torch.floor(scan / 0.4).long(); k2 = (key[:,0]+4096) + ((key[:,1]+4096) << 14) + ((key[:,2]+4096) << 28)
scan_s = scan[torch.argsort(k2)].contiguous()
t_knn_s = ev(lambda: ops.knn_query(st, scan_s, 8, pose=np.eye(4), out=out))
bricks = ops.BrickCache(dx, 2)
t_build = ev(lambda: bricks.build(st), n=5)
print("bricks", bricks.n_bricks, "entries", bricks.n_entries, "build ms", t_build)
t_knn_b = ev(lambda: ops.knn_query(st, scan, 8, pose=np.eye(4), out=out, bricks=bricks))
t_knn_bs = ev(lambda: ops.knn_query(st, scan_s, 8, pose=np.eye(4), out=out, bricks=bricks))
print(f"knn bricks {t_knn_b*1e3:.1f} us  bricks(sorted) {t_knn_bs*1e3:.1f} us")
print(f"layers={layers} P={P} knn {t_knn*1e3:.1f} us  knn(sorted) {t_knn_s*1e3:.1f} us  gn {t_gn*1e3:.1f} us  sdf_query {t_sdf*1e3:.1f} us")
s = sums.cpu().numpy().sum(0); print("valid", s[29])
