"""Pin the oracle's restatement of the configuration branches of the `variants` / `process_sem` fixtures (recorded from the
unmodified reference by oracle/make_golden.py gen_variants / gen_process_sem on the maps of c2_wf / kitti_nwf):
reg_dist_div_grad_norm, and the semantic head -- Decoder.sem_label_prob, Tracker.query_source_points(query_sem), two
Mapper.mapping iterations with the NLL term, Mapper.process_frame with per-point labels.  CPU only."""
import numpy as np
import pytest

from oracle import pin_oracle as O
from tests import golden_util as G


@pytest.fixture(scope="module", params=["c2_wf", "kitti_nwf"])
def vg(request):
    d = G.load(request.param)
    v = G.load("variants")
    d["table"] = G.dense_table(d)
    d["name"] = request.param
    d["v"] = {k[len(request.param) + 1:]: x for k, x in v.items() if k.startswith(request.param + "_")}
    d["params"] = O.unpack_decoder(d["dec_flat"], 11, int(d["dec_hidden"]), int(d["dec_levels"]))
    S = int(d["v"]["sem_heads"])
    d["sparams"] = O.unpack_decoder(d["v"]["sem_dec_flat"], 11, int(d["dec_hidden"]), int(d["dec_levels"]), out_dim=S)
    return d


def _search(d, q, tf=True):
    return O.radius_search(q, d["table"], d["neural_points"], d["resolution"], d["neighbor_dx"], d["max_valid_dist2"],
                           ts_create=d["point_ts_create"], travel_dist=d["travel_dist"] if tf else None, cur_ts=int(d["cur_ts"]),
                           diff_travel_dist_local=d["diff_travel_dist_local"])


def test_registration_step_with_the_residual_divided_by_the_gradient_norm(vg):
    d, v = vg, vg["v"]
    k = int(d["query_nn_k"])
    s = _search(d, d["reg_cur"])
    sdf, grad, std, nn, _ = O.query_sdf(d["reg_cur"], s, d["local_geo_features"], d["local_neural_points"], d["params"], d["sdf_scale"], k,
                                        weighted_first=bool(d["weighted_first"]), global2local=d["global2local"])
    kw = dict(valid_nn_k=int(d["track_mask_query_nn_k"]), min_grad_norm=d["cfg_reg_min_grad_norm"], max_grad_norm=d["cfg_reg_max_grad_norm"],
              max_sdf_std=d["cfg_surface_sample_range_m"] * d["cfg_max_sdf_std_ratio"], GM_dist=d["cfg_reg_GM_dist_m"],
              GM_grad=d["cfg_reg_GM_grad"], lm_lambda=d["cfg_reg_lm_lambda"], sdf_labels=np.zeros(len(sdf)))
    r = O.registration_step(d["reg_cur"], sdf, grad, std, nn, dist_div_grad_norm=True, **kw)
    assert r["valid_count"] == int(v["ddgn_valid_count"])
    np.testing.assert_allclose(r["T"], v["ddgn_dT"], rtol=0, atol=3e-6)
    assert abs(r["residual_cm"] - v["ddgn_residual_cm"]) < 1e-4 * max(1.0, v["ddgn_residual_cm"])
    plain = O.registration_step(d["reg_cur"], sdf, grad, std, nn, **kw)
    assert np.abs(plain["T"] - r["T"]).max() > 1e-6  # (the switch matters on this input)


def test_sem_label_prob_and_query(vg):
    d, v = vg, vg["v"]
    k, wf = int(d["query_nn_k"]), bool(d["weighted_first"])
    s = _search(d, d["query"])
    qf = O.query_feature(d["query"], s, d["local_geo_features"], d["local_neural_points"], None, k, global2local=d["global2local"],
                         weighted_first=wf)
    lp = O.sem_label_prob(qf["geo_feat"].astype(np.float64), tuple([w.astype(np.float64) for w in p] if isinstance(p, list) else p.astype(np.float64) for p in d["sparams"]))
    np.testing.assert_allclose(lp, v["sem_prob"], rtol=2e-5, atol=2e-5)
    assert np.allclose(np.exp(lp).sum(-1), 1.0, atol=1e-9)
    pred, label, nn = O.query_sem(d["query"], s, d["local_geo_features"], d["local_neural_points"], d["sparams"], k, weighted_first=wf,
                                  global2local=d["global2local"])
    ref = v["sem_pred"].astype(np.int64)
    # an argmax can flip where the two best classes are closer than the arithmetic noise: none on these fixtures
    top2 = np.sort(pred, -1)[:, -2:]
    clear = (top2[:, 1] - top2[:, 0]) > 1e-5
    assert clear.mean() > 0.99 and np.array_equal(label[clear], ref[clear])
    assert len(np.unique(ref)) > 5  # (not a degenerate head)


def test_mapping_two_iterations_with_the_semantic_term(vg):
    """Gradients of the geometry features (SDF term + semantic term), of the SDF decoder and of the semantic decoder, the NLL
    term and the total loss of both iterations, the parameters after the two Adam steps."""
    d, v = vg, vg["v"]
    k, wf = int(d["query_nn_k"]), bool(d["weighted_first"])
    S = int(v["sem_heads"])
    feats = d["local_geo_features"].astype(np.float64).copy()
    flat, sflat = d["dec_flat"].astype(np.float64).copy(), v["sem_dec_flat"].astype(np.float64).copy()
    cert, tsu = d["local_point_certainties"].copy(), d["local_point_ts_update"].copy()
    mf, vf = np.zeros_like(feats), np.zeros_like(feats)
    md, vd, ms, vs = np.zeros_like(flat), np.zeros_like(flat), np.zeros_like(sflat), np.zeros_like(sflat)
    shape = (11, int(d["dec_hidden"]), int(d["dec_levels"]))
    lr, aeps = float(v["cfg_lr"]), float(v["cfg_adam_eps"])
    gfs, gds, gss = [], [], []
    for it in range(2):
        coord = v[f"map_coord{it}"]

        def searcher(points, main=[True]):
            s = _search(d, points)
            train = main[0]
            main[0] = False
            qf = O.query_feature(points, s, feats.astype(np.float32), d["local_neural_points"], cert, k, global2local=d["global2local"],
                                 weighted_first=False, training_mode=train, query_ts=v[f"map_ts{it}"] if train else None, ts_update=tsu)
            if train:
                searcher.side = (qf["certainties_after"], qf["ts_update_after"])
            return qf

        r = O.train_step(coord, v[f"map_label{it}"], np.ones(len(coord), np.float32), searcher, feats, d["local_neural_points"], flat, shape,
                         d["sdf_scale"], k, weighted_first=wf, dec=int(v["cfg_gradient_decimation"]), eps=v["map_eps"],
                         weight_e=v["cfg_weight_e"])
        cert, tsu = searcher.side

        def plain(points):
            return O.query_feature(points, _search(d, points), feats.astype(np.float32), d["local_neural_points"], None, k,
                                   global2local=d["global2local"], weighted_first=False)

        rs = O.train_sem_step(coord, v[f"map_sem{it}"], plain, feats, sflat, shape + (S,), k, weighted_first=wf, weight_s=v["cfg_weight_s"],
                              decimation=int(v["cfg_sem_label_decimation"]), freespace_label_on=bool(v["cfg_freespace_label_on"]))
        gfeat = r["feat_grad"] + rs["feat_grad"]
        gf, gd, gs = v[f"map_gfeat{it}"], v[f"map_gdec{it}"], v[f"map_gsem{it}"]
        assert np.max(np.abs(gfeat - gf)) < 2e-4 * np.abs(gf).max()
        assert np.max(np.abs(r["dec_grad"] - gd)) < 2e-4 * np.abs(gd).max()
        assert np.max(np.abs(rs["dec_grad"] - gs)) < 2e-4 * np.abs(gs).max()
        assert abs(rs["loss"] - v["map_loss_sem"][it]) < 1e-5 * abs(v["map_loss_sem"][it])
        assert abs(r["sdf_loss"] - v["map_loss_sdf"][it]) < 1e-5 * abs(v["map_loss_sdf"][it])
        total = r["loss"] + v["cfg_weight_s"] * rs["loss"]
        assert abs(total - v["map_loss_total"][it]) < 1e-5 * abs(v["map_loss_total"][it])
        assert 0 < rs["selected"].sum() < len(coord)
        gfs.append(gfeat); gds.append(r["dec_grad"]); gss.append(rs["dec_grad"])
        feats, mf, vf = O.adam_step(feats, gfeat, mf, vf, it + 1, lr, eps=aeps)
        flat, md, vd = O.adam_step(flat, r["dec_grad"], md, vd, it + 1, lr, eps=aeps)
        sflat, ms, vs = O.adam_step(sflat, rs["dec_grad"], ms, vs, it + 1, lr, eps=aeps)
    G.adam_outliers(feats, v["map_feat_after"], gfs, [v["map_gfeat0"], v["map_gfeat1"]], lr)
    G.adam_outliers(flat, v["map_dec_after"], gds, [v["map_gdec0"], v["map_gdec1"]], lr)
    G.adam_outliers(sflat, v["map_sem_after"], gss, [v["map_gsem0"], v["map_gsem1"]], lr)


def test_process_frame_carries_the_semantic_labels():
    """sem_label_pool after every frame: the sampler's labels (point label for the measured point and its close-to-surface
    samples, 0 for free space), appended, then filtered by the pool's window / discard mask."""
    d = G.load("process_sem")
    Sn, Fn, Bn = int(d["surface_sample_n"]), int(d["free_front_n"]), int(d["free_behind_n"])
    pool_sem, pool_glob = np.zeros((0,), np.int32), np.zeros((0, 3), np.float32)
    for t in range(int(d["n_frames"])):
        f = f"f{t}_"
        scan = d[f + "scan"]
        coord, label, _, _ = O.sample_rays(scan, None, d[f + "rnd_surface"], d[f + "rnd_front"], d[f + "rnd_behind"],
                                           surface_range=d["surface_sample_range_m"], surface_n=Sn, front_n=Fn, behind_n=Bn,
                                           free_begin_ratio=d["free_sample_begin_ratio"], free_end_dist=d["free_sample_end_dist_m"],
                                           dist_weight_on=bool(d["dist_weight_on"]), dist_weight_scale=d["dist_weight_scale"],
                                           max_range=d["max_range"], behind_dropoff_on=bool(d["behind_dropoff_on"]))
        sem = O.sample_sem_labels(d[f + "labels"], Sn, Fn, Bn)
        assert len(sem) == len(label)
        assert np.array_equal(sem != 0, (np.repeat(d[f + "labels"], 1 + Sn + Fn + Bn) != 0) & (np.tile(np.arange(1 + Sn + Fn + Bn), len(scan)) <= Sn))
        pool_sem = np.concatenate([pool_sem, sem])
        pool_glob = np.concatenate([pool_glob, O.transform_points(coord, d[f + "pose"])])
        mask = O.pool_filter_mask(pool_glob, d[f + "pose"][:3, 3], float(d["window_radius"]), pool_capacity=int(d["pool_capacity"]),
                                  discard_index=d[f + "discard_index"])
        pool_sem, pool_glob = pool_sem[mask], pool_glob[mask]
        assert np.array_equal(pool_sem, d[f + "after_sem_label_pool"]), t
        assert len(pool_sem) == int(d[f + "pool_sample_count"])
