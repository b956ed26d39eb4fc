import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pin_slam_amd import ops, synth, _lib
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
feats = dev(m.features); cert = torch.zeros(P, device="cuda")
N = 100_000
scan = dev(synth.make_scan(m, n=N))
key = torch.floor(scan / 0.4).long()
scan = scan[torch.argsort((key[:, 0] + 4096) + ((key[:, 1] + 4096) << 14) + ((key[:, 2] + 4096) << 28))].contiguous()
bricks = ops.BrickCache(dx, 2).build(st, wait=True)
L = _lib.lib()
state = torch.zeros(64, dtype=torch.float64, device="cuda")
T0 = np.eye(4)
stream = torch.cuda.current_stream().cuda_stream
_lib.check(L.pin_gn_state_init(state.data_ptr(), T0.ctypes.data, N, stream), "init")
sums = torch.zeros((64, 32), dtype=torch.float64, device="cuda")
nn = torch.empty(N, dtype=torch.int32, device="cuda")
for (H, Lv) in ((64, 4), (64, 1)):
    fs = ops.FieldState(feats=feats, dec=dev(synth.init_decoder(H, Lv)), k=8, hidden=H, levels=Lv, weighted_first=True,
                        sdf_scale=0.055, certainty=cert, pos=pos)
    sp, f, bc = st.params(time_filtering=True, local=True), fs.params(), bricks.params()
    for _ in range(40):
        _lib.check(L.pin_gn_iteration(C.byref(sp), C.byref(bc), C.byref(f), C.byref(gp), scan.data_ptr(), N, 8, None,
                                      sums.data_ptr(), state.data_ptr(), None, None, nn.data_ptr(), stream), "it")
    torch.cuda.synchronize()
print("done")
