"""GPU parity of the Mapper.process_frame data path kernels (K12-K14) through the C ABI against
the fixtures recorded from the reference (tests/golden/process*.npz) and the oracle."""
import numpy as np
import pytest
import torch

from oracle import pin_oracle as O
from tests import golden_util as G
from tests.test_oracle_process import POOLS, frame_pool_before

pytestmark = pytest.mark.gpu


class _Cfg:
    pass


def _cfg(d):
    c = _Cfg()
    for k in ("surface_sample_range_m", "surface_sample_n", "free_front_n", "free_behind_n", "free_sample_begin_ratio",
              "free_sample_end_dist_m", "dist_weight_on", "dist_weight_scale", "max_range", "behind_dropoff_on"):
        setattr(c, k, d[k])
    return c


@pytest.fixture(scope="module", params=["process", "process_color"])
def pg(request):
    return G.load(request.param)


def _names(d):
    return POOLS + (("color_pool",) if "f0_s_color" in d else ())


FIELD_OF = dict(coord_pool="coord", global_coord_pool="global_coord", sdf_label_pool="sdf_label", weight_pool="weight",
                time_pool="ts", color_pool="color")


def test_sampler_and_pool_filter_follow_reference(pg):
    """Four frames of append -> window/discard/compaction; every pool array after every frame
    equals the reference's (bit-exact except the transformed coordinates: sgemm vs fma order)."""
    from pin_slam_amd import pool as P
    d = pg
    C = 3 if "f0_s_color" in d else 0
    pool = P.SamplePool(color_channels=C, capacity=1024)
    for t in range(int(d["n_frames"])):
        f = f"f{t}_"
        scan = torch.from_numpy(d[f + "scan"]).cuda()
        rnd = tuple(torch.from_numpy(d[f + k]).cuda() for k in ("rnd_surface", "rnd_front", "rnd_behind"))
        n0 = pool.n
        n_new = pool.append_samples(scan, P.sample_params(_cfg(d), d[f + "pose"], t), rnd=rnd)
        assert n_new == len(d[f + "s_label"])
        tail = {k: pool.view(FIELD_OF[k])[n0:].cpu().numpy() for k in _names(d)}
        assert np.array_equal(tail["coord_pool"].view(np.uint32), d[f + "s_coord"].view(np.uint32))
        assert np.array_equal(tail["sdf_label_pool"].view(np.uint32), d[f + "s_label"].view(np.uint32))
        assert np.array_equal(tail["weight_pool"].view(np.uint32), d[f + "s_weight"].view(np.uint32))
        assert (tail["time_pool"] == t).all()
        if C:
            assert np.array_equal(tail["color_pool"], d[f + "s_color"])
        np.testing.assert_allclose(tail["global_coord_pool"], O.transform_points(d[f + "s_coord"], d[f + "pose"]),
                                   rtol=0, atol=4e-6)
        disc = torch.from_numpy(d[f + "discard_index"]).cuda()
        n_pool, n_cur = pool.filter(d[f + "pose"][:3, 3], d["window_radius"], int(d["pool_capacity"]),
                                    discard_index=disc if disc.numel() else None)
        assert (n_pool, n_cur) == (d[f + "pool_sample_count"], d[f + "cur_sample_count"])
        for k in _names(d):
            got, ref = pool.view(FIELD_OF[k]).cpu().numpy(), d[f + "after_" + k]
            if k == "global_coord_pool":
                np.testing.assert_allclose(got, ref, rtol=0, atol=4e-6)
            else:
                assert np.array_equal(got, ref), (k, t)


def test_query_certainty_and_new_index(pg):
    from pin_slam_amd import ops
    d = pg
    B = int(d["buffer_size"])
    dx, mv = ops.search_neighborhood(1, 0.0, d["resolution"])
    cand = torch.from_numpy(ops.candidate_offsets(dx, B)).cuda()
    for t in range(int(d["n_frames"])):
        f = f"f{t}_"
        table = np.full(B, -1, np.int32)
        table[d[f + "qc_table_slots"]] = d[f + "qc_table_vals"]
        pos = torch.from_numpy(d[f + "qc_positions"]).cuda()
        P_ = pos.shape[0]
        pos4 = torch.zeros((P_, 4), dtype=torch.float32, device="cuda")
        ops.pack_positions(pos, torch.zeros(P_, dtype=torch.int32, device="cuda"), pos4, 0, P_)
        st = ops.SearchState(table=torch.from_numpy(table).cuda(), pos4=pos4, cand_off=cand, n_points=P_,
                             resolution=d["resolution"], max_valid_dist2=mv)
        cur = int(d[f + "cur_sample_count"])
        q = torch.from_numpy(d[f + "after_global_coord_pool"][-cur:].copy()).cuda()
        cert = ops.query_certainty(st, torch.from_numpy(d[f + "qc_certainties"]).cuda(), q)
        assert np.array_equal(cert.cpu().numpy(), d[f + "qc_out"])
        lab = torch.from_numpy(d[f + "after_sdf_label_pool"][-cur:].copy()).cuda()
        idx, cnt = ops.new_sample_index(cert, lab, d["new_certainty_thre"], d["surface_sample_range_m"] * 3.0,
                                        offset=int(d[f + "pool_sample_count"]) - cur)
        c = int(cnt.item())
        assert np.array_equal(idx[:c].cpu().numpy(), d[f + "new_idx"])


def test_sampler_large_random_matches_oracle():
    """100k-point scan with torch-drawn noise (the product path's RNG use): bit-exact vs the oracle."""
    from pin_slam_amd import pool as P
    from pin_slam_amd.config import PinConfig
    torch.manual_seed(3)
    cfg = PinConfig()
    N = 100_000
    scan = (torch.rand(N, 3, device="cuda") - 0.5) * torch.tensor([80.0, 80.0, 6.0], device="cuda")
    pose = np.eye(4); pose[:3, 3] = [1.0, -2.0, 0.5]
    g = torch.Generator(device="cuda").manual_seed(5)
    rnd = (torch.randn(N * 3, 1, device="cuda", generator=g), torch.rand(N * 2, 1, device="cuda", generator=g),
           torch.rand(N * 1, 1, device="cuda", generator=g))
    pool = P.SamplePool(capacity=1024)
    pool.append_samples(scan, P.sample_params(cfg, pose, 4), rnd=rnd)
    coord, label, _, weight = O.sample_rays(scan.cpu().numpy(), None, *(r.cpu().numpy().reshape(-1) for r in rnd),
                                            surface_range=cfg.surface_sample_range_m, surface_n=3, front_n=2, behind_n=1,
                                            free_begin_ratio=cfg.free_sample_begin_ratio, free_end_dist=cfg.free_sample_end_dist_m,
                                            dist_weight_on=cfg.dist_weight_on, dist_weight_scale=cfg.dist_weight_scale,
                                            max_range=cfg.max_range)
    assert np.array_equal(pool.view("coord").cpu().numpy().view(np.uint32), coord.view(np.uint32))
    assert np.array_equal(pool.view("sdf_label").cpu().numpy().view(np.uint32), label.view(np.uint32))
    assert np.array_equal(pool.view("weight").cpu().numpy().view(np.uint32), weight.view(np.uint32))


class _Spy:
    """Records torch.randn / rand / randint results (the drop-in draws them in the reference's order)."""

    def __enter__(self):
        self.calls = []
        self._orig = {n: getattr(torch, n) for n in ("randn", "rand", "randint")}
        for n, f in self._orig.items():
            def wrap(*a, _f=f, _n=n, **k):
                r = _f(*a, **k)
                self.calls.append((_n, r.detach().cpu().numpy().copy()))
                return r
            setattr(torch, n, wrap)
        return self

    def __exit__(self, *exc):
        for n, f in self._orig.items():
            setattr(torch, n, f)


class _Dataset:
    lose_track = False
    stop_status = False
    static_mask = None

    def __init__(self, n):
        self.processed_frame = 0
        self.odom_poses = np.tile(np.eye(4), (n, 1, 1))
        self.pgo_poses = self.gt_poses = self.odom_poses
        self.gt_pose_provided = True


def test_dropin_process_frame_matches_oracle(monkeypatch):
    """Mapper.process_frame through the drop-in classes for four frames (with Mapper.mapping in
    between so certainties are real), replayed step by step with the oracle on the same random
    draws: sample pools, counts, new-sample index, adaptive offset and the grown map agree."""
    from pin_slam_amd import pool as P
    from pin_slam_amd.config import PinConfig
    from pin_slam_amd.dropin.model.decoder import Decoder
    from pin_slam_amd.dropin.model.neural_points import NeuralPoints
    from pin_slam_amd.dropin.utils.mapper import Mapper
    d = G.load("process")
    B = 40009
    cfg = PinConfig(voxel_size_m=0.4, buffer_size=B, local_map_radius=22.0, local_map_travel_dist_ratio=5.0, bs=1024,
                    bs_new_sample=256, feature_std=0.05, pool_capacity=33000, pool_filter_freq=1, window_radius=13.0,
                    max_range=20.0, adaptive_iters=True, search_alpha=0.5, query_nn_k=6)
    torch.manual_seed(0)
    npts = NeuralPoints(cfg)
    nfr = int(d["n_frames"])
    ds = _Dataset(nfr)
    mp = Mapper(cfg, ds, npts, {"sdf": Decoder(cfg, 32, 1, 1), "semantic": None, "color": None})
    snap = {}
    orig_filter = P.SamplePool.filter_begin

    def spy_filter(self, *a, **k):
        snap["global"] = self.view("global_coord").cpu().numpy().copy()
        return orig_filter(self, *a, **k)

    monkeypatch.setattr(P.SamplePool, "filter_begin", spy_filter)
    state = dict(table=np.full(B, -1, np.int64), positions=np.zeros((0, 3), np.float32),
                 ts_create=np.zeros(0, np.int32), ts_update=np.zeros(0, np.int32))
    pools = {k: np.zeros((0, 3) if "coord" in k else (0,), np.int32 if k == "time_pool" else np.float32) for k in POOLS}
    kw = dict(surface_range=cfg.surface_sample_range_m, surface_n=3, front_n=2, behind_n=1,
              free_begin_ratio=cfg.free_sample_begin_ratio, free_end_dist=cfg.free_sample_end_dist_m,
              dist_weight_on=True, dist_weight_scale=cfg.dist_weight_scale, max_range=cfg.max_range)
    travel = [0.0]
    discards = 0
    for t in range(nfr):
        pose = d[f"f{t}_pose"]
        if t:
            travel.append(travel[-1] + float(np.linalg.norm(pose[:3, 3] - d[f"f{t-1}_pose"][:3, 3])))
        ds.odom_poses[t] = pose
        ds.processed_frame = t
        npts.travel_dist = torch.tensor(travel + [0.0] * (nfr - len(travel)), dtype=torch.float32, device="cuda")
        scan = torch.from_numpy(d[f"f{t}_scan"]).cuda()
        with _Spy() as spy:
            mp.process_frame(scan, None, torch.tensor(pose, dtype=torch.float64, device="cuda"), t)
        names = [c[0] for c in spy.calls]
        assert names[:4] == ["randn", "rand", "rand", "randn"], names  # surface, front, behind, new-point features
        # ---- oracle replay on the same draws
        coord, label, _, weight = O.sample_rays(d[f"f{t}_scan"], None, *(c[1].reshape(-1) for c in spy.calls[:3]), **kw)
        g_or = O.transform_points(coord, pose)
        n_new = len(label)
        g_gpu = snap["global"]
        np.testing.assert_allclose(g_gpu[-n_new:], g_or, rtol=0, atol=4e-6)
        upd = g_gpu[-n_new:][np.abs(label) < np.float32(cfg.surface_sample_range_m * cfg.map_surface_ratio)]
        O.map_update(state, upd, t, cfg.voxel_size_m, travel_dist=np.asarray(travel + [0.0] * (nfr - len(travel)), np.float32),
                     diff_travel_dist_local=npts.diff_travel_dist_local)
        assert npts.count() == len(state["positions"])
        assert np.array_equal(npts.neural_points.cpu().numpy(), state["positions"])
        # 40 009 slots: many samples of one call collide; the table keeps the LAST sample's value like the reference
        assert np.array_equal(npts.buffer_pt_index.cpu().numpy().astype(np.int64), state["table"])
        before = dict(coord_pool=np.concatenate([pools["coord_pool"], coord]), global_coord_pool=g_gpu,
                      sdf_label_pool=np.concatenate([pools["sdf_label_pool"], label]),
                      weight_pool=np.concatenate([pools["weight_pool"], weight]),
                      time_pool=np.concatenate([pools["time_pool"], np.full(n_new, t, np.int32)]))
        disc = [c[1] for c in spy.calls if c[0] == "randint"]
        discards += len(disc)
        mask = O.pool_filter_mask(g_gpu, pose[:3, 3], cfg.window_radius, cfg.pool_capacity, disc[0] if disc else None)
        pools = {k: v[mask] for k, v in before.items()}
        assert mp.pool_sample_count == int(mask.sum()) and mp.cur_sample_count == int(mask[-n_new:].sum())
        for k in POOLS:
            assert np.array_equal(getattr(mp, k).cpu().numpy(), pools[k]), (k, t)
        cur = mp.cur_sample_count
        cert = O.query_certainty(pools["global_coord_pool"][-cur:], state["table"], state["positions"],
                                 npts.point_certainties.cpu().numpy(), cfg.voxel_size_m)
        idx = O.new_sample_index(cert, pools["sdf_label_pool"][-cur:], cfg.new_certainty_thre, cfg.surface_sample_range_m,
                                 offset=mp.pool_sample_count - cur)
        got = mp.new_idx.cpu().numpy()
        assert np.array_equal(got, idx), (t, len(got), len(idx), np.setdiff1d(got, idx)[:5], np.setdiff1d(idx, got)[:5])
        assert mp.adaptive_iter_offset == O.adaptive_iter_offset(len(idx), cur, t, adaptive_iters=True)
        # ---- get_batch: the reference's draw order (history, then new samples), rows gathered from the pool
        with _Spy() as spy:
            co, lab, ts, _, _, col, w = mp.get_batch(global_coord=True)
        hist, newb = spy.calls[0][1], spy.calls[1][1]
        index = np.concatenate([hist, idx[newb]])
        assert len(index) == cfg.bs and len(newb) == min(len(idx), cfg.bs_new_sample)
        assert np.array_equal(co.cpu().numpy(), pools["global_coord_pool"][index])
        assert np.array_equal(lab.cpu().numpy(), pools["sdf_label_pool"][index])
        assert np.array_equal(ts.cpu().numpy(), pools["time_pool"][index])
        assert np.array_equal(w.cpu().numpy(), pools["weight_pool"][index])
        mp.mapping(6)
        state["ts_update"] = npts.point_ts_update.cpu().numpy().copy()  # training refreshes ts_update of touched points
    assert discards > 0
    assert float(npts.point_certainties.max()) > 0
