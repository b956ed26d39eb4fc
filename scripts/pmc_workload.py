"""Workload for the PMC passes: the tracker's kNN launch and the fused GN launch on the bench's
C3 map, plus a calibration copy of known size (MI355X_MICROARCH.md: FETCH_SIZE on gfx950 has to
be calibrated against a known byte count).  Run under `rocprofv3 --pmc <counter>`."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pin_slam_amd import ops, synth
from pin_slam_amd._lib import GnParams

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 16
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
key = torch.floor(scan / 0.4).long()
scan = scan[torch.argsort((key[:, 0] + 4096) + ((key[:, 1] + 4096) << 14) + ((key[:, 2] + 4096) << 28))].contiguous()
gp = GnParams(); gp.valid_nn_k = 8; gp.min_grad_norm = 0.5; gp.max_grad_norm = 2.0; gp.max_sdf_std = 0.25; gp.gm_dist = 0.3; gp.gm_grad = 0.1
# calibration: copy of 1 GiB (reads 1 GiB, writes 1 GiB), well past the 256 MiB Infinity Cache
a = torch.empty(1 << 28, dtype=torch.float32, device="cuda").normal_()
b = torch.empty_like(a)
torch.cuda.synchronize()
for _ in range(3):
    b.copy_(a)
torch.cuda.synchronize()
out = None
bricks = ops.BrickCache(dx, 2).build(st, wait=True)
for i in range(10):
    nbr, nn, cur = ops.knn_query(st, scan, 8, pose=np.eye(4), out=out)       # direct hash probe (r01 a-c)
    nbr, nn, cur = ops.knn_query(st, scan, 8, pose=np.eye(4), out=out, bricks=bricks)  # brick cache
    out = (nbr, nn, cur)
    sums, _, _ = ops.gn_accumulate(fs, gp, cur, nbr, nn)
torch.cuda.synchronize()
print("done", P)
