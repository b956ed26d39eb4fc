"""Edge cases of the hot path through the C ABI on MI355X: empty inputs, ragged sizes around the
tile / wave / block boundaries, queries with no neighbour at all, a one-point scan -- against the oracle
or by consistency between sizes (every kernel is per-query, so a prefix must give a prefix)."""
import dataclasses

import numpy as np
import pytest
import torch

from oracle import pin_oracle as O
from tests import golden_util as G

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gd():
    from tests import gpu_util as U
    d = G.load("c2_wf")
    d["st"], d["fs"] = U.search_state(d), U.field_state(d)
    d["table"] = G.dense_table(d)
    return d


def _gp(d):
    from pin_slam_amd._lib import GnParams
    gp = GnParams()
    gp.valid_nn_k = int(d["track_mask_query_nn_k"]); gp.min_grad_norm = 0.3; gp.max_grad_norm = 3.0
    gp.max_sdf_std = 0.5; gp.gm_dist = 0.3; gp.gm_grad = 0.1
    return gp


def test_empty_inputs_are_no_ops(gd):
    from pin_slam_amd import ops, pool as P, preprocess as PP
    from pin_slam_amd.config import PinConfig
    d = gd
    k = int(d["query_nn_k"])
    q0 = torch.empty((0, 3), dtype=torch.float32, device="cuda")
    nbr, nn, _ = ops.knn_query(d["st"], q0, k)
    assert nbr.shape == (0, k, 4) and nn.shape == (0,)
    sdf, grad, std, cert = ops.sdf_query(d["fs"], q0, nbr, nn)
    assert sdf.shape == (0,) and grad.shape == (0, 3)
    pool = P.SamplePool(capacity=1024)
    assert pool.filter(np.zeros(3), 10.0, 100) == (0, 0)
    n = pool.append_samples(torch.empty((0, 3), dtype=torch.float32, device="cuda"), P.sample_params(PinConfig(), np.eye(4), 0))
    assert n == 0 and pool.n == 0
    pts, ts = PP.crop_frame(torch.empty((0, 4), dtype=torch.float32, device="cuda"), None)
    assert pts.shape == (0, 4) and ts is None


@pytest.mark.parametrize("n", [1, 15, 16, 17, 63, 64, 65, 255, 257, 1000])
def test_ragged_sizes_give_prefixes(gd, n):
    """kNN, SDF query and the GN sums at sizes around the 16-query tile, the 64-lane wave and the 256-thread
    block: the first n results never depend on what comes after them."""
    from pin_slam_amd import ops
    d = gd
    k = int(d["query_nn_k"])
    q_all = torch.from_numpy(d["query"]).cuda()
    nbr_all, nn_all, _ = ops.knn_query(d["st"], q_all, k)
    sdf_all, grad_all, _, _ = ops.sdf_query(d["fs"], q_all, nbr_all, nn_all)
    q = q_all[:n].contiguous()
    nbr, nn, _ = ops.knn_query(d["st"], q, k)
    assert torch.equal(nbr.view(torch.int32), nbr_all[:n].view(torch.int32)) and torch.equal(nn, nn_all[:n])
    sdf, grad, _, _ = ops.sdf_query(d["fs"], q, nbr, nn)
    assert torch.equal(sdf, sdf_all[:n]) and torch.equal(grad, grad_all[:n])
    # GN sums of the prefix: the per-point outputs of the fused kernel are the SDF query's, the count is exact
    sums, s2, g2 = ops.gn_accumulate(d["fs"], _gp(d), q, nbr, nn, want_points=True)
    np.testing.assert_allclose(s2.cpu().numpy(), sdf.cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(g2.cpu().numpy(), grad.cpu().numpy(), rtol=1e-4, atol=1e-5)
    tot = sums.cpu().numpy().sum(0)
    gn = np.linalg.norm(grad.cpu().numpy(), axis=1)
    valid = (nn.cpu().numpy() >= int(d["track_mask_query_nn_k"])) & (gn < 3.0) & (gn > 0.3)
    assert int(round(tot[29])) == int(valid.sum())


def test_queries_without_any_neighbour(gd):
    """Points far outside the map (incl. > 2^29 cells away): no candidates, zero features, mask off, no NaN."""
    from pin_slam_amd import ops
    d = gd
    k = int(d["query_nn_k"])
    far = np.array([[500.0, -300.0, 40.0], [1e6, 1e6, -1e6], [3e8, 0.0, 0.0], [-3e8, 5.0, 2.0]], np.float32)
    q = torch.from_numpy(np.concatenate([far, d["query"][:12]])).cuda()
    bricks = ops.BrickCache(d["neighbor_dx"], int(d["num_nei_cells"])).build(d["st"], wait=True)
    for b in (None, bricks):
        nbr, nn, _ = ops.knn_query(d["st"], q, k, bricks=b)
        assert (nn[:4] == 0).all() and (nbr[:4, :, 3].view(torch.int32) == -1).all()
        s = O.radius_search(q.cpu().numpy(), d["table"], d["neural_points"], d["resolution"], d["neighbor_dx"], d["max_valid_dist2"],
                            ts_create=d["point_ts_create"], travel_dist=d["travel_dist"], cur_ts=int(d["cur_ts"]),
                            diff_travel_dist_local=d["diff_travel_dist_local"])
        assert np.array_equal(nn.cpu().numpy(), (s[1] >= 0).sum(1))
        sdf, grad, std, cert = ops.sdf_query(d["fs"], q, nbr, nn)
        assert torch.isfinite(sdf).all() and torch.isfinite(grad).all() and (cert[:4] == 0).all()
        sums, _, _ = ops.gn_accumulate(d["fs"], _gp(d), q, nbr, nn)
        assert np.isfinite(sums.cpu().numpy()).all()


def test_tiny_training_batch_matches_oracle(gd):
    """A 20-sample batch (2 Eikonal samples, 32 queries: half a wave) through pin_train_step vs the oracle."""
    from pin_slam_amd import ops
    from tests import gpu_util as U
    d = gd
    k, H, L = int(d["query_nn_k"]), int(d["dec_hidden"]), int(d["dec_levels"])
    bs, dec = 20, int(d["map_dec"])
    coord, label, w, ts = d["map_coord0"][:bs], d["map_label0"][:bs], d["map_w0"][:bs], d["map_ts0"][:bs]
    feats, decf = U.dev(d["local_geo_features"]), U.dev(d["dec_flat"])
    cert, tsu = U.dev(d["local_point_certainties"]), U.dev(d["local_point_ts_update"], torch.int32)
    fs = dataclasses.replace(d["fs"], feats=feats, dec=decf, certainty=cert)
    buf = ops.TrainBuffers(bs, dec, k, H, L)
    gfeat, gdec = torch.zeros_like(feats), torch.zeros_like(decf)
    ops.train_step(d["st"], fs, buf, U.dev(coord), U.dev(label), U.dev(w), U.dev(ts, torch.int32), cert, tsu, gfeat, gdec,
                   sigma=d["sdf_scale"], weight_e=d["map_weight_e"], eik_eps=d["map_eps"])

    def searcher(p):
        s = O.radius_search(p, d["table"], d["neural_points"], d["resolution"], d["neighbor_dx"], d["max_valid_dist2"],
                            ts_create=d["point_ts_create"], travel_dist=d["travel_dist"], cur_ts=int(d["cur_ts"]),
                            diff_travel_dist_local=d["diff_travel_dist_local"])
        return O.query_feature(p, s, d["local_geo_features"], d["local_neural_points"], None, k, global2local=d["global2local"],
                               weighted_first=False)
    ref = O.train_step(coord, label, w, searcher, d["local_geo_features"], d["local_neural_points"], d["dec_flat"], (11, H, L),
                       d["sdf_scale"], k, dec=dec, eps=d["map_eps"], weight_e=d["map_weight_e"])
    gf, gd_ = ref["feat_grad"], ref["dec_grad"]
    assert np.max(np.abs(gfeat.cpu().numpy() - gf)) < 3e-4 * np.abs(gf).max()
    assert np.max(np.abs(gdec.cpu().numpy() - gd_)) < 3e-4 * np.abs(gd_).max()


@pytest.fixture(scope="module")
def gd_nwf():
    """kitti_nwf: weighted_first = False, decoder 1 x 64, k = 6 -- the per-neighbour fused training kernel."""
    from tests import gpu_util as U
    d = G.load("kitti_nwf")
    d["st"], d["fs"] = U.search_state(d), U.field_state(d)
    d["table"] = G.dense_table(d)
    return d


@pytest.mark.parametrize("mode", ["wf", "nwf"])
@pytest.mark.parametrize("bs,dec,eikonal", [(37, 3, True), (5, 1, True), (64, 1000, True), (33, 10, False), (212, 7, True)])
def test_fused_training_tile_map(gd, gd_nwf, mode, bs, dec, eikonal):
    """Tile -> query map of the fused training kernels (train_fused.h).  Interpolate-first: mixed tiles of 2 x 6 probes +
    4 main samples; per-neighbour decoding: groups of three (query, neighbour) tiles, six main samples or the six probes
    of an Eikonal sample.  Odd Eikonal counts, more probes than main samples to fill tiles (dec = 1), one Eikonal sample,
    Eikonal off, ragged last groups, a frozen decoder -- feature / decoder gradients and both loss sums vs the oracle."""
    from pin_slam_amd import ops
    from tests import gpu_util as U
    d = gd if mode == "wf" else gd_nwf
    wf = mode == "wf"
    assert bool(d["fs"].weighted_first) == wf
    k, H, L = int(d["query_nn_k"]), int(d["dec_hidden"]), int(d["dec_levels"])
    coord, label, w, ts = d["map_coord0"][:bs], d["map_label0"][:bs], d["map_w0"][:bs], d["map_ts0"][:bs]
    assert len(coord) == bs
    feats, decf = U.dev(d["local_geo_features"]), U.dev(d["dec_flat"])
    fs = dataclasses.replace(d["fs"], feats=feats, dec=decf, certainty=U.dev(d["local_point_certainties"]))
    buf = ops.TrainBuffers(bs, dec, k, H, L, eikonal=eikonal, weighted_first=wf)
    assert buf.n_eik == (len(range(0, bs, dec)) if eikonal else 0)
    kw = dict(sigma=d["sdf_scale"], weight_e=d["map_weight_e"], eik_eps=d["map_eps"])

    def run(with_dec):
        cert, tsu = U.dev(d["local_point_certainties"]), U.dev(d["local_point_ts_update"], torch.int32)
        gfeat, gdec = torch.zeros_like(feats), (torch.zeros_like(decf) if with_dec else None)
        pred = torch.empty(bs, device="cuda")
        loss = ops.train_step(d["st"], fs, buf, U.dev(coord), U.dev(label), U.dev(w), U.dev(ts, torch.int32), cert, tsu, gfeat,
                              gdec, pred_out=pred, **kw).cpu().numpy().copy()
        return gfeat.cpu().numpy(), None if gdec is None else gdec.cpu().numpy(), loss, pred.cpu().numpy(), cert.cpu().numpy()

    def searcher(p):
        s = O.radius_search(p, d["table"], d["neural_points"], d["resolution"], d["neighbor_dx"], d["max_valid_dist2"],
                            ts_create=d["point_ts_create"], travel_dist=d["travel_dist"], cur_ts=int(d["cur_ts"]),
                            diff_travel_dist_local=d["diff_travel_dist_local"])
        return O.query_feature(p, s, d["local_geo_features"], d["local_neural_points"], None, k, global2local=d["global2local"],
                               weighted_first=False)
    ref = O.train_step(coord, label, w, searcher, d["local_geo_features"], d["local_neural_points"], d["dec_flat"], (11, H, L),
                       d["sdf_scale"], k, dec=dec, eps=d["map_eps"], weight_e=d["map_weight_e"], ekional=eikonal, weighted_first=wf)
    gfeat, gdec, loss, pred, cert = run(True)
    gf, gd_ = ref["feat_grad"], ref["dec_grad"]
    assert np.max(np.abs(gfeat - gf)) < 1e-4 * np.abs(gf).max()
    assert np.max(np.abs(gdec - gd_)) < 1e-4 * np.abs(gd_).max()
    np.testing.assert_allclose(pred, ref["sdf_pred"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(loss[0] / bs, ref["sdf_loss"], rtol=1e-5)
    if eikonal:
        np.testing.assert_allclose(loss[1] / buf.n_eik, ref["eik_loss"], rtol=2e-4)
    else:
        assert loss[1] == 0.0
    gfeat2, none, loss2, pred2, cert2 = run(False)  # frozen decoder: same feature gradient (atomics: not bit for bit), same side effects
    assert none is None
    assert np.max(np.abs(gfeat2 - gfeat)) < 1e-6 * np.abs(gf).max()
    np.testing.assert_allclose(loss2, loss, rtol=1e-12)
    assert np.array_equal(pred2, pred) and np.allclose(cert2, cert, rtol=1e-6, atol=1e-7)


def test_one_point_scan_through_the_sampler_and_voxel_filter():
    from pin_slam_amd import pool as P, preprocess as PP
    from pin_slam_amd.config import PinConfig
    cfg = PinConfig()
    scan = torch.tensor([[3.0, -4.0, 1.0]], device="cuda")
    pool = P.SamplePool(capacity=64)
    g = torch.Generator(device="cuda").manual_seed(1)
    rnd = (torch.randn(3, 1, device="cuda", generator=g), torch.rand(2, 1, device="cuda", generator=g), torch.rand(1, 1, device="cuda", generator=g))
    assert pool.append_samples(scan, P.sample_params(cfg, np.eye(4), 0), rnd=rnd) == 7
    coord, label, _, weight = O.sample_rays(scan.cpu().numpy(), None, *(r.cpu().numpy().reshape(-1) for r in rnd),
                                            surface_range=cfg.surface_sample_range_m, surface_n=3, front_n=2, behind_n=1,
                                            free_begin_ratio=cfg.free_sample_begin_ratio, free_end_dist=cfg.free_sample_end_dist_m,
                                            max_range=cfg.max_range)
    assert np.array_equal(pool.view("coord").cpu().numpy().view(np.uint32), coord.view(np.uint32))
    assert np.array_equal(pool.view("sdf_label").cpu().numpy().view(np.uint32), label.view(np.uint32))
    idx = PP.voxel_down_sample_torch(scan, 0.4)
    assert idx.tolist() == [0]
    kept, _ = PP.crop_frame(scan, None, min_range=10.0)  # 5.1 m < 10 m: cropped away
    assert kept.shape[0] == 0


@pytest.mark.parametrize("n", [1, 2, 63, 1000])
def test_spatial_sort_small_and_degenerate_inputs(n):
    """pin_spatial_sort on tiny inputs, duplicated points, negative coordinates and points beyond the 1024-cell wrap:
    always a permutation of the input, keys ascending."""
    from pin_slam_amd import ops
    rng = np.random.default_rng(n)
    p = (rng.standard_normal((n, 3)) * 40.0).astype(np.float32)
    if n >= 63:
        p[10:20] = p[5]            # duplicates
        p[20] = (900.0, -700.0, 3.0)  # beyond +-512 cells of 0.5 m: the code wraps
    out, perm = ops.spatial_sort(torch.from_numpy(p).cuda(), 0.5, return_perm=True)
    pm = perm.cpu().numpy()
    assert np.array_equal(np.sort(pm), np.arange(n))
    assert np.array_equal(out.cpu().numpy(), p[pm])
    cell = (np.floor(out.cpu().numpy() / np.float32(0.5)).astype(np.int64) + 512) & 1023

    def spread(v):
        r = np.zeros_like(v)
        for b in range(10):
            r |= ((v >> b) & 1) << (3 * b)
        return r
    key = spread(cell[:, 0]) | (spread(cell[:, 1]) << 1) | (spread(cell[:, 2]) << 2)
    assert (np.diff(key) >= 0).all()
