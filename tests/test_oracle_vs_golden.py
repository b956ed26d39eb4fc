"""Pin the numpy oracle (oracle/pin_oracle.py) against fixtures produced by the real,
unmodified reference (oracle/make_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import pin_oracle as O
from tests import golden_util as G


@pytest.fixture(scope="module", params=G.CASES)
def gold(request):
    d = G.load(request.param)
    d["table"] = G.dense_table(d)
    d["params"] = O.unpack_decoder(d["dec_flat"], 11, int(d["dec_hidden"]), int(d["dec_levels"]))
    return d


def _search(d, q, tf):
    kw = {}
    if tf:
        kw = dict(ts_create=d["point_ts_create"], travel_dist=d["travel_dist"], cur_ts=int(d["cur_ts"]),
                  diff_travel_dist_local=d["diff_travel_dist_local"])
    return O.radius_search(q, d["table"], d["neural_points"], d["resolution"], d["neighbor_dx"],
                           d["max_valid_dist2"], **kw)


def test_search_neighborhood(gold):
    dx, mv = O.search_neighborhood(int(gold["num_nei_cells"]), gold["search_alpha"], gold["resolution"])
    assert np.array_equal(dx, gold["neighbor_dx"])
    assert mv == gold["max_valid_dist2"]


@pytest.mark.parametrize("tf", [0, 1])
def test_radius_search_bit_exact(gold, tf):
    d2, idx = _search(gold, gold["query"], tf)
    assert np.array_equal(idx, gold[f"rs_idx_tf{tf}"])
    assert np.array_equal(d2.view(np.uint32), gold[f"rs_d2_tf{tf}"].view(np.uint32))
    assert (idx >= 0).sum() > 1000  # the fixture is not vacuous


@pytest.mark.parametrize("tag", ["loc", "glob"])
def test_query_feature(gold, tag):
    d, q = gold, gold["query"]
    loc = tag == "loc"
    s = _search(d, q, tf=loc)
    if loc:
        qf = O.query_feature(q, s, d["local_geo_features"], d["local_neural_points"],
                             d["local_point_certainties"], int(d["query_nn_k"]),
                             global2local=d["global2local"], weighted_first=bool(d["weighted_first"]))
    else:
        qf = O.query_feature(q, s, d["geo_features"], d["neural_points"], d["point_certainties"],
                             int(d["query_nn_k"]), weighted_first=bool(d["weighted_first"]))
    assert np.array_equal(qf["nn_count"], d[f"qf_{tag}_nn"])
    np.testing.assert_allclose(qf["weight"], d[f"qf_{tag}_w"], rtol=2e-6, atol=1e-9)
    np.testing.assert_allclose(qf["geo_feat"], d[f"qf_{tag}_feat"], rtol=1e-5, atol=2e-7)
    np.testing.assert_allclose(qf["certainty"], d[f"qf_{tag}_cert"], rtol=1e-5, atol=1e-6)


def _qsdf(d, q):
    s = _search(d, q, tf=True)
    return O.query_sdf(q, s, d["local_geo_features"], d["local_neural_points"], d["params"],
                       d["sdf_scale"], int(d["query_nn_k"]), weighted_first=bool(d["weighted_first"]),
                       global2local=d["global2local"], certainties=d["local_point_certainties"])


def test_query_source_points(gold):
    d = gold
    sdf, grad, std, nn, cert = _qsdf(d, d["query"])
    assert np.array_equal(nn >= d["track_mask_query_nn_k"], d["qsp_mask"])
    np.testing.assert_allclose(sdf, d["qsp_sdf"], rtol=1e-5, atol=1e-7)
    # the reference's own float32 autograd gradient carries ~5e-5 relative noise
    scale = np.abs(d["qsp_grad"]).max(1, keepdims=True) + 1e-6
    assert np.max(np.abs(grad - d["qsp_grad"]) / scale) < 1e-4
    np.testing.assert_allclose(std, d["qsp_std"], rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(cert, d["qsp_cert"], rtol=1e-5, atol=1e-6)


def test_after_pgo(gold):
    """after_pgo: apply_quaternion_rotation (utils/tools.py:428-437) on the neighbour vectors."""
    d = gold
    s = _search(d, d["query"], tf=True)
    sdf, grad, std, nn, _ = O.query_sdf(d["query"], s, d["local_geo_features"], d["local_neural_points"], d["params"],
                                        d["sdf_scale"], int(d["query_nn_k"]), weighted_first=bool(d["weighted_first"]),
                                        global2local=d["global2local"], orientations=d["pgo_quat"])
    np.testing.assert_allclose(sdf, d["pgo_sdf"], rtol=1e-5, atol=1e-7)
    scale = np.abs(d["pgo_grad"]).max(1, keepdims=True) + 1e-6
    assert np.max(np.abs(grad - d["pgo_grad"]) / scale) < 1e-4
    qf = O.query_feature(d["query"], s, d["local_geo_features"], d["local_neural_points"], None, int(d["query_nn_k"]),
                         global2local=d["global2local"], orientations=d["pgo_quat"],
                         weighted_first=bool(d["weighted_first"]))
    np.testing.assert_allclose(qf["geo_feat"], d["pgo_feat"], rtol=1e-5, atol=3e-7)


def _reg_kwargs(d):
    return dict(valid_nn_k=int(d["track_mask_query_nn_k"]), min_grad_norm=d["cfg_reg_min_grad_norm"],
                max_grad_norm=d["cfg_reg_max_grad_norm"],
                max_sdf_std=d["cfg_surface_sample_range_m"] * d["cfg_max_sdf_std_ratio"],
                GM_dist=d["cfg_reg_GM_dist_m"], GM_grad=d["cfg_reg_GM_grad"], lm_lambda=d["cfg_reg_lm_lambda"])


def test_transform_and_registration_step(gold):
    d = gold
    cur = O.transform_points(d["reg_src"], d["reg_Tinit"])
    np.testing.assert_allclose(cur, d["reg_cur"], rtol=0, atol=2e-6)
    sdf, grad, std, nn, _ = _qsdf(d, d["reg_cur"])
    r = O.registration_step(d["reg_cur"], sdf, grad, std, nn, **_reg_kwargs(d))
    assert r["valid_count"] == d["reg_valid_count"]
    assert abs(r["residual_cm"] - d["reg_residual_cm"]) < 1e-3 * max(1.0, d["reg_residual_cm"])
    np.testing.assert_allclose(r["T"], d["reg_dT"], rtol=0, atol=2e-6)


def test_tracking_loop(gold):
    """Tracker.tracking (utils/tracker.py:114-184): GN loop, same termination rule."""
    d = gold
    T = d["reg_Tinit"].copy()
    iter_n = int(d["cfg_reg_iter_n"])
    converged = False
    for i in range(iter_n):
        cur = O.transform_points(d["reg_src"], T)
        sdf, grad, std, nn, _ = _qsdf(d, cur)
        r = O.registration_step(cur, sdf, grad, std, nn, **_reg_kwargs(d))
        T = r["T"] @ T
        if converged:
            break
        dT = r["T"]
        ang = np.degrees(np.arccos(np.clip((np.trace(dT[:3, :3]) - 1) / 2, -1, 1)))
        if (abs(ang) < d["cfg_reg_term_thre_deg"] and np.linalg.norm(dT[:3, 3]) < d["cfg_reg_term_thre_m"]) \
                or i == iter_n - 2:
            converged = True
    assert d["trk_valid"]
    np.testing.assert_allclose(T[:3, 3], d["trk_T"][:3, 3], rtol=0, atol=1e-4)
    np.testing.assert_allclose(T[:3, :3], d["trk_T"][:3, :3], rtol=0, atol=1e-5)


def test_mapping_two_iterations(gold):
    """Mapper.mapping on fixed batches: gradients of iteration 0/1, post-Adam parameters,
    certainty / ts side effects (mapper.py:645-818, neural_points.py:685-710)."""
    d = gold
    k = int(d["query_nn_k"])
    wf = bool(d["weighted_first"])
    feats = d["local_geo_features"].astype(np.float64).copy()
    flat = d["dec_flat"].astype(np.float64).copy()
    cert = d["local_point_certainties"].copy()
    tsu = d["local_point_ts_update"].copy()
    mf, vf = np.zeros_like(feats), np.zeros_like(feats)
    md, vd = np.zeros_like(flat), np.zeros_like(flat)
    shape = (11, int(d["dec_hidden"]), int(d["dec_levels"]))
    gfs, gds = [], []
    for it in range(2):
        coord = d[f"map_coord{it}"]

        def searcher(points, main=[True]):
            s = _search(d, points, tf=True)
            train = main[0]
            main[0] = False
            qf = O.query_feature(points, s, feats.astype(np.float32), d["local_neural_points"], cert, k,
                                 global2local=d["global2local"], weighted_first=False,
                                 training_mode=train, query_ts=d[f"map_ts{it}"] if train else None,
                                 ts_update=tsu)
            if train:
                searcher.side = (qf["certainties_after"], qf["ts_update_after"])
            return qf

        r = O.train_step(coord, d[f"map_label{it}"], d[f"map_w{it}"], searcher, feats, d["local_neural_points"],
                         flat, shape, d["sdf_scale"], k, weighted_first=wf, dec=int(d["map_dec"]),
                         eps=d["map_eps"], weight_e=d["map_weight_e"], loss_weight_on=bool(d["map_loss_weight_on"]))
        cert, tsu = searcher.side
        gf, gd = d[f"map_gfeat{it}"], d[f"map_gdec{it}"]
        assert np.max(np.abs(r["feat_grad"] - gf)) < 2e-4 * np.abs(gf).max()
        assert np.max(np.abs(r["dec_grad"] - gd)) < 2e-4 * np.abs(gd).max()
        # the scalar losses of the reference (BCE term, and the total at its backward() call)
        assert abs(r["sdf_loss"] - d["map_loss_sdf"][it]) < 1e-5 * abs(d["map_loss_sdf"][it])
        assert abs(r["loss"] - d["map_loss_total"][it]) < 1e-5 * abs(d["map_loss_total"][it])
        gfs.append(r["feat_grad"]); gds.append(r["dec_grad"])
        feats, mf, vf = O.adam_step(feats, r["feat_grad"], mf, vf, it + 1, d["map_lr"], eps=d["map_adam_eps"])
        flat, md, vd = O.adam_step(flat, r["dec_grad"], md, vd, it + 1, d["map_lr"], eps=d["map_adam_eps"])
    # post-Adam parameters: tight wherever the gradient is above its measured rounding noise (golden_util.adam_outliers)
    frac_f, _ = G.adam_outliers(feats, d["map_feat_after"], gfs, [d["map_gfeat0"], d["map_gfeat1"]], d["map_lr"])
    frac_d, _ = G.adam_outliers(flat, d["map_dec_after"], gds, [d["map_gdec0"], d["map_gdec1"]], d["map_lr"])
    print(d.get("name"), "noise-dominated entries: features", frac_f, "decoder", frac_d)
    assert frac_f < 0.2 and frac_d < 0.2
    np.testing.assert_allclose(cert, d["map_cert_after"], rtol=1e-4, atol=1e-5)
    assert np.array_equal(tsu, d["map_ts_after"])


def test_update_and_local_map():
    """NeuralPoints.update / reset_local_map (neural_points.py:311-513) including the
    voxel down-sampler (utils/tools.py:583-626)."""
    d = G.load("update")
    B = int(d["buffer_size"])
    st = dict(table=np.full(B, -1, np.int64), positions=np.zeros((0, 3), np.float32),
              ts_create=np.zeros(0, np.int32), ts_update=np.zeros(0, np.int32))
    for ts in range(4):
        pts = d[f"pts{ts}"]
        assert np.array_equal(O.voxel_down_sample(pts, d["resolution"]), d[f"sel{ts}"])
        O.map_update(st, pts, ts, d["resolution"], travel_dist=d["travel_dist"],
                     diff_travel_dist_local=d["diff_travel_dist_local"])
        assert st["positions"].shape[0] == d[f"count{ts}"]
        mask, g2l = O.local_map_mask(st["positions"], st["ts_create"], [9.0 * ts, 0, 0], d["local_map_radius"],
                                     travel_dist=d["travel_dist"], cur_ts=ts,
                                     diff_travel_dist_local=d["diff_travel_dist_local"], reboot_ts=0)
        assert np.array_equal(mask, d[f"local_mask{ts}"][:-1])
        assert np.array_equal(g2l, d[f"global2local{ts}"])
    assert np.array_equal(st["positions"], d["neural_points"])
    assert np.array_equal(st["ts_create"], d["point_ts_create"])
    slots = np.nonzero(st["table"] >= 0)[0]
    assert np.array_equal(slots, d["table_slots"])
    assert np.array_equal(st["table"][slots], d["table_vals"])


def local_map_variants(d):
    """(name, oracle arguments) of the reset_local_map variants recorded in the `update` fixture."""
    P, tc, tu = d["neural_points"], d["point_ts_create"], d["point_ts_update"]
    sp64, sp32 = d["var_sensor"], d["var_sensor"].astype(np.float32)
    td = dict(travel_dist=d["travel_dist"], diff_travel_dist_local=d["diff_travel_dist_local"])
    mid = O.mid_ts(tc, tu)
    return [("var_ts_mask", (P, tc, sp32, d["local_map_radius"]), dict(cur_ts=2, diff_ts_local=2)),
            ("var_f64_mask", (P, tc, sp64, d["local_map_radius"]), dict(cur_ts=3, **td)),
            ("var_f32_mask", (P, tc, sp32, d["local_map_radius"]), dict(cur_ts=3, **td)),
            ("var_mid_mask", (P, mid, sp32, d["local_map_radius"]), dict(cur_ts=3, **td)),
            ("var_mid_ts_mask", (P, mid, sp32, d["local_map_radius"]), dict(cur_ts=2, diff_ts_local=1))]


def test_local_map_variants():
    """reset_local_map by a window of frames (use_travel_dist=False, pin_slam.py:287), with a float64 sensor position,
    with config.use_mid_ts."""
    d = G.load("update")
    seen = []
    for name, a, kw in local_map_variants(d):
        mask, g2l = O.local_map_mask(*a, **kw)
        assert np.array_equal(mask, d[name][:-1]), name
        seen.append(mask)
        if name == "var_ts_mask":
            assert np.array_equal(g2l, d["var_ts_g2l"])
    assert not np.array_equal(seen[0], seen[4]) and not np.array_equal(seen[2], seen[3])  # the variants matter


# ---------------------------------------------------------------- colour path (C5, run_replica.yaml)
@pytest.fixture(scope="module")
def cgold():
    d = G.load("replica_color")
    d["table"] = G.dense_table(d)
    d["params"] = O.unpack_decoder(d["dec_flat"], 11, 64, 1)
    d["cparams"] = O.unpack_decoder(d["cdec_flat"], 11, 64, 1, out_dim=3)
    return d


def _csearch(d, q):
    return O.radius_search(q, d["table"], d["neural_points"], d["resolution"], d["neighbor_dx"], d["max_valid_dist2"],
                           ts_create=d["point_ts_create"], travel_dist=d["travel_dist"], cur_ts=int(d["cur_ts"]),
                           diff_travel_dist_local=d["diff_travel_dist_local"])


def test_color_query(cgold):
    d = cgold
    s = _csearch(d, d["query"])
    col, cgrad, nn = O.query_color(d["query"], s, d["local_color_features"], d["local_neural_points"], d["cparams"],
                                   int(d["query_nn_k"]), global2local=d["global2local"])
    np.testing.assert_allclose(col, d["qsp_color"], rtol=1e-5, atol=1e-6)
    scale = np.abs(d["qsp_color_grad"]).max(-1, keepdims=True) + 1e-5
    assert np.max(np.abs(cgrad - d["qsp_color_grad"]) / scale) < 2e-4
    qf = O.query_feature(d["query"], s, d["local_color_features"], d["local_neural_points"], None, int(d["query_nn_k"]),
                         global2local=d["global2local"])
    np.testing.assert_allclose(qf["geo_feat"], d["qf_color_feat"], rtol=1e-5, atol=3e-7)


@pytest.mark.parametrize("tag", ["photo", "consist"])
def test_color_registration_step(cgold, tag):
    d = cgold
    k = int(d["query_nn_k"])
    s = _csearch(d, d["reg_cur"])
    sdf, grad, std, nn, _ = O.query_sdf(d["reg_cur"], s, d["local_geo_features"], d["local_neural_points"], d["params"],
                                        d["sdf_scale"], k, global2local=d["global2local"])
    col, cgrad, _ = O.query_color(d["reg_cur"], s, d["local_color_features"], d["local_neural_points"], d["cparams"], k,
                                  global2local=d["global2local"])
    r = O.registration_step(d["reg_cur"], sdf, grad, std, nn, valid_nn_k=int(d["track_mask_query_nn_k"]),
                            min_grad_norm=d["cfg_reg_min_grad_norm"], max_grad_norm=d["cfg_reg_max_grad_norm"],
                            max_sdf_std=d["surface_sample_range_m"] * d["cfg_max_sdf_std_ratio"],
                            GM_dist=d["cfg_reg_GM_dist_m"], GM_grad=d["cfg_reg_GM_grad"], lm_lambda=d["cfg_reg_lm_lambda"],
                            colors=d["reg_colors"], color_pred=col, color_grad=cgrad, photo_loss=(tag == "photo"),
                            photo_weight=d["photometric_loss_weight"])
    assert r["valid_count"] == d[f"reg_valid_{tag}"]
    np.testing.assert_allclose(r["T"], d[f"reg_dT_{tag}"], rtol=0, atol=3e-6)
    if tag == "photo":
        assert abs(r["photo_residual"] - d["reg_photo_res_photo"]) < 1e-5


def test_color_mapping_gradients(cgold):
    """Iteration 0 of Mapper.mapping with colour: gradients of geo/colour features and of both
    decoders against the reference's autograd."""
    d = cgold
    k = int(d["query_nn_k"])

    def searcher_for(feats):
        def searcher(points):
            return O.query_feature(points, _csearch(d, points), feats, d["local_neural_points"], None, k,
                                   global2local=d["global2local"], weighted_first=False)
        return searcher

    r = O.train_step(d["map_coord0"], d["map_label0"], d["map_w0"], searcher_for(d["local_geo_features"]),
                     d["local_geo_features"], d["local_neural_points"], d["dec_flat"], (11, 64, 1), d["sdf_scale"], k,
                     dec=int(d["map_dec"]), eps=d["map_eps"], weight_e=d["map_weight_e"])
    rc = O.train_color_step(d["map_coord0"], d["map_label0"], d["map_color0"], d["map_w0"],
                            searcher_for(d["local_color_features"]), d["local_color_features"], d["cdec_flat"],
                            (11, 64, 1, 3), k, surface_range=d["surface_sample_range_m"], weight_i=d["weight_i"])
    for got, ref in ((r["feat_grad"], d["map_gfeat0"]), (r["dec_grad"], d["map_gdec0"]),
                     (rc["feat_grad"], d["map_cfeat0"]), (rc["dec_grad"], d["map_cdec0"])):
        assert np.max(np.abs(got - ref)) < 3e-4 * np.abs(ref).max()


@pytest.mark.parametrize("tag", ["nwf", "pgo", "wf"])
def test_analytic_eikonal_mapping(tag):
    """Mapper.mapping with numerical_grad_on False (run_livox.yaml:27): the Eikonal term on the autograd gradient of every
    sample, differentiated a second time (mapper.py:642-643, 677-678, 760-782).  Per-neighbour decoding, the same with
    rotated neighbour vectors (after a pose-graph correction), and weighted-first decoding: gradients and scalar losses
    of every iteration against the reference's double backward."""
    d = G.load("analytic_eik")
    d["table"] = G.dense_table(d)
    k, wf = int(d["query_nn_k"]), tag == "wf"
    orient = d["pgo_quat"] if tag in ("pgo", "wf") else None  # (the wf run comes after the pgo run in the generator)
    if tag == "wf":
        orient = None  # after_pgo was switched off again
    feats = d[f"{tag}_feat_before"].astype(np.float64).copy()
    flat = d[f"{tag}_dec_before"].astype(np.float64).copy()
    cert, tsu = d[f"{tag}_cert_before"].copy(), d[f"{tag}_tsu_before"].copy()
    mf, vf, md, vd = np.zeros_like(feats), np.zeros_like(feats), np.zeros_like(flat), np.zeros_like(flat)
    shape = (11, int(d["dec_hidden"]), int(d["dec_levels"]))
    n_it = len(d[f"{tag}_loss_total"])
    for it in range(n_it):
        def searcher(points):
            s = _search(d, points, tf=True)
            return O.query_feature(points, s, feats.astype(np.float32), d["local_neural_points"], cert, k,
                                   global2local=d["global2local"], orientations=orient, weighted_first=False,
                                   training_mode=True, query_ts=d[f"{tag}_ts{it}"], ts_update=tsu)

        r = O.train_step(d[f"{tag}_coord{it}"], d[f"{tag}_label{it}"], d[f"{tag}_w{it}"], searcher, feats,
                         d["local_neural_points"], flat, shape, d["sdf_scale"], k, weighted_first=wf, dec=1,
                         weight_e=d[f"{tag}_weight_e"], loss_weight_on=bool(d[f"{tag}_loss_weight_on"]), analytic=True,
                         orientations=orient)
        gf, gd = d[f"{tag}_gfeat{it}"], d[f"{tag}_gdec{it}"]
        assert np.max(np.abs(r["feat_grad"] - gf)) < 2e-4 * np.abs(gf).max()
        assert np.max(np.abs(r["dec_grad"] - gd)) < 2e-4 * np.abs(gd).max()
        assert r["eik_loss"] > 0.01  # the term is live
        assert abs(r["sdf_loss"] - d[f"{tag}_loss_sdf"][it]) < 1e-5 * abs(d[f"{tag}_loss_sdf"][it])
        assert abs(r["loss"] - d[f"{tag}_loss_total"][it]) < 1e-5 * abs(d[f"{tag}_loss_total"][it])
        cert, tsu = r["fw"]["qf"]["certainties_after"], r["fw"]["qf"]["ts_update_after"]
        feats, mf, vf = O.adam_step(feats, r["feat_grad"], mf, vf, it + 1, d[f"{tag}_lr"], eps=d[f"{tag}_adam_eps"])
        flat, md, vd = O.adam_step(flat, r["dec_grad"], md, vd, it + 1, d[f"{tag}_lr"], eps=d[f"{tag}_adam_eps"])
    np.testing.assert_allclose(cert, d[f"{tag}_cert_after"], rtol=1e-4, atol=1e-5)
    assert np.array_equal(tsu, d[f"{tag}_ts_after"])
