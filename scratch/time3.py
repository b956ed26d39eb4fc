import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pin_slam_amd import ops, synth, engine
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
H, L = 64, 4
fs = ops.FieldState(feats=dev(m.features), dec=dev(synth.init_decoder(H, L)), k=8, hidden=H, levels=L, weighted_first=True,
                    sdf_scale=0.055, certainty=torch.zeros(P, device="cuda"), pos=pos)
pc, pl = synth.make_pool(m, n=200_000)
bs = 16384
coord, label = dev(pc[:bs]), dev(pl[:bs])
w = torch.ones(bs, device="cuda"); ts = torch.zeros(bs, dtype=torch.int32, device="cuda")
tsu = torch.zeros(P, dtype=torch.int32, device="cuda")
buf = ops.TrainBuffers(bs, 10, 8, H, L)
gfeat = torch.zeros_like(fs.feats); gdec = torch.zeros_like(fs.dec)
bricks = ops.BrickCache(dx, 2).build(st)
mode = sys.argv[1] if len(sys.argv) > 1 else "full"
for i in range(30):
    ops.train_step(st, fs, buf, coord, label, w, ts, fs.certainty, tsu, gfeat, gdec if mode != "nodec" else None,
                   sigma=0.055, weight_e=0.5, eik_eps=0.08, bricks=bricks)
torch.cuda.synchronize(); print("done", mode)
