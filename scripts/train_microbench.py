"""One Mapper.mapping training step (pin_train_step: fused tile kernel + streamed weight gradient + finalize) timed alone
on the bench's C3 map for several batch sizes, with and without the decoder gradient (HIP events over `reps` calls; the
kNN records are computed once).  `PIN_LIB=<path>` loads another build of libpinhip.so for A/B runs."""
import os, sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pin_slam_amd._lib as _L  # noqa: E402
_L.LIB_PATH = os.environ.get("PIN_LIB", _L.LIB_PATH)
from pin_slam_amd import ops, synth  # noqa: E402
import ctypes as C  # noqa: E402

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 16
sizes = [int(a) for a in sys.argv[2].split(",")] if len(sys.argv) > 2 else [16384, 131072, 1 << 20]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
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
bricks = ops.BrickCache(dx, 2).build(st, wait=True)
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
print("lib", _L.LIB_PATH)
for bs in sizes:
    idx = torch.randint(0, scan.shape[0], (bs,), device="cuda", generator=gen)
    coord = (scan[idx] + 0.05 * torch.randn((bs, 3), device="cuda", generator=gen)).contiguous()
    if os.environ.get("SORT_BATCH"):  # probe: a spatially ordered batch (what sorting the drawn pool indices would give)
        kk = (torch.floor(coord / 0.1).long() + 512) & 1023
        def spread(v):
            v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3
            return (v | (v << 2)) & 0x09249249
        coord = coord[torch.argsort(spread(kk[:, 0]) | (spread(kk[:, 1]) << 1) | (spread(kk[:, 2]) << 2))].contiguous()
    label = (0.05 * torch.randn((bs,), device="cuda", generator=gen)).contiguous()
    w = torch.ones(bs, device="cuda"); ts = torch.zeros(bs, dtype=torch.int32, device="cuda")
    buf = ops.TrainBuffers(bs, 10, 8, H, L)
    gfeat, gdec = torch.zeros_like(fs.feats), torch.zeros_like(fs.dec)
    tsu = torch.zeros(P, dtype=torch.int32, device="cuda")
    kw = dict(sigma=0.055, weight_e=0.5, eik_eps=0.02)
    ops.train_step(st, fs, buf, coord, label, w, ts, fs.certainty, tsu, gfeat, gdec, bricks=bricks, **kw)  # queries + kNN
    tp = _L.TrainParams()
    tp.n_main, tp.n_eik, tp.loss_weight_on = buf.n_main, buf.n_eik, 0
    tp.sigma, tp.weight_e, tp.eik_eps = 0.055, 0.5, float(np.float32(0.02))
    tp.inv_n_main, tp.inv_n_eik = 1.0 / buf.n_main, 1.0 / max(buf.n_eik, 1)
    f = fs.params()
    P_ = ops._ptr

    def step(dec_grad):
        ops.check(_L.lib().pin_train_step(C.byref(f), C.byref(tp), P_(buf.query), P_(buf.nbr), P_(buf.nn), P_(label), P_(w), P_(ts),
                                          *((None, None) if os.environ.get("NO_SIDE") else (P_(fs.certainty), P_(tsu))), P_(gfeat), P_(dec_grad), P_(buf.loss), None, P_(buf.ws),
                                          buf.ws.numel() * 4, ops._stream()), "pin_train_step")

    def timeit(fn, n=reps):
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

    t_knn = timeit(lambda: ops.knn_query(st, buf.query, 8, out=(buf.nbr, buf.nn, None), bricks=bricks))
    t_full, t_frozen = timeit(lambda: step(gdec)), timeit(lambda: step(None))
    print(f"bs={bs:8d} queries={buf.Q:8d}  train_step {t_full:8.1f} us ({bs / t_full:7.1f} samples/us)   frozen decoder {t_frozen:8.1f} us   kNN {t_knn:6.1f} us"
          f"   loss {buf.loss.cpu().numpy()}")

    if os.environ.get("TF_STAMPS"):  # debug build (scripts/build_variant.sh NAME -DPIN_TF_STAMPS=<wave> [-DPIN_TF_STAMP_TILE=<n>]): phases of that tile
        buf_s = (C.c_ulonglong * (1024 * 16))()
        assert C.CDLL(_L.LIB_PATH).pin_debug_tf_stamps(buf_s, 1024 * 16) == 0
        a = np.frombuffer(buf_s, dtype=np.uint64).reshape(1024, 16).astype(np.int64)[:256, :8]
        a = a[(a[:, 7] > a[:, 0]) & (a[:, 0] > 0)]
        names = ["top -> loads requested", "-> image in LDS (barrier)", "-> gather arithmetic done", "-> forward layers", "-> head + loss",
                 "-> backward sweep", "-> scatter issued"]
        d = np.diff(a, axis=1) * 10.0 / 1e3
        if len(a) == 0:
            print("  no stamped tiles at this size"); continue
        print(f"  stamped tiles: {len(a)}; whole tile {np.mean((a[:, 7] - a[:, 0]) * 0.01):.2f} us (median {np.median((a[:, 7] - a[:, 0]) * 0.01):.2f})")
        for i, n in enumerate(names):
            print(f"    {n:32s} mean {d[:, i].mean():6.2f} us   median {np.median(d[:, i]):6.2f}   p90 {np.percentile(d[:, i], 90):6.2f}")
