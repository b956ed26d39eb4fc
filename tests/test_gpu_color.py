"""Colour path (C5, run_replica.yaml) on the GPU against the reference's recorded outputs."""
import dataclasses

import numpy as np
import pytest
import torch

from tests import golden_util as G

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cg():
    from pin_slam_amd import ops
    from tests import gpu_util as U
    d = G.load("replica_color")
    d["st"] = U.search_state(d)
    d["fs"] = U.field_state(d, local=True)
    d["fc"] = dataclasses.replace(d["fs"], feats=U.dev(d["local_color_features"]), dec=U.dev(d["cdec_flat"]),
                                  hidden=int(d["cdec_hidden"]), levels=int(d["cdec_levels"]), out_dim=3)
    return d


def test_color_query_and_channel_gradients(cg):
    from pin_slam_amd import ops
    from tests import gpu_util as U
    d = cg
    q = U.dev(d["query"])
    nbr, nn, _ = ops.knn_query(d["st"], q, int(d["query_nn_k"]))
    feat, _, _ = ops.query_feature(d["fc"], q, nbr, nn)
    np.testing.assert_allclose(feat.cpu().numpy(), d["qf_color_feat"], rtol=1e-5, atol=3e-7)
    np.testing.assert_allclose(ops.decoder_color(d["fc"], feat).cpu().numpy(), d["qsp_color"], rtol=1e-4, atol=2e-6)
    for c in range(3):
        kap = [0.0, 0.0, 0.0]; kap[c] = 1.0
        col, val, g = ops.color_query(d["fc"], q, nbr, nn, kappa=kap)
        np.testing.assert_allclose(col.cpu().numpy(), d["qsp_color"], rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(val.cpu().numpy(), d["qsp_color"][:, c], rtol=1e-4, atol=2e-6)
        ref = d["qsp_color_grad"][:, c, :]
        scale = np.abs(ref).max(1, keepdims=True) + 1e-5
        assert np.max(np.abs(g.cpu().numpy() - ref) / scale) < 3e-4
    col, val, g = ops.color_query(d["fc"], q, nbr, nn)  # intensity
    ref = np.einsum("c,ncj->nj", np.array(ops.INTENSITY), d["qsp_color_grad"])
    assert np.max(np.abs(g.cpu().numpy() - ref) / (np.abs(ref).max(1, keepdims=True) + 1e-5)) < 3e-4


def _gp(d):
    from pin_slam_amd._lib import GnParams
    gp = GnParams()
    gp.valid_nn_k = int(d["track_mask_query_nn_k"])
    gp.min_grad_norm, gp.max_grad_norm = d["cfg_reg_min_grad_norm"], d["cfg_reg_max_grad_norm"]
    gp.max_sdf_std = d["surface_sample_range_m"] * d["cfg_max_sdf_std_ratio"]
    gp.gm_dist, gp.gm_grad = d["cfg_reg_GM_dist_m"], d["cfg_reg_GM_grad"]
    return gp


@pytest.mark.parametrize("tag", ["photo", "consist"])
def test_color_registration_step(cg, tag):
    """registration_step with colours: photometric term (implicit_color_reg) or the colour
    consistency weight (utils/tracker.py:493-542, 699-744)."""
    from pin_slam_amd import ops
    from tests import gpu_util as U
    d = cg
    src = U.dev(d["reg_src"])
    nbr, nn, cur = ops.knn_query(d["st"], src, int(d["query_nn_k"]), pose=d["reg_Tinit"])
    ct, keep = ops.color_term(d["fc"], U.dev(d["reg_colors"]), photometric=(tag == "photo"),
                              photo_weight=d["photometric_loss_weight"])
    sums, _, _ = ops.gn_accumulate(d["fs"], _gp(d), cur, nbr, nn, color=ct)
    T, cnt, res_cm, extra = ops.solve_gn(sums.cpu().numpy(), d["cfg_reg_lm_lambda"])
    assert abs(cnt - d[f"reg_valid_{tag}"]) <= 2
    np.testing.assert_allclose(T, d[f"reg_dT_{tag}"], rtol=0, atol=1e-5)
    if tag == "photo":
        assert abs(extra["photo_residual"] - d["reg_photo_res_photo"]) < 1e-4


def test_color_tracking(cg):
    from pin_slam_amd import engine, ops
    from tests import gpu_util as U
    d = cg
    src = U.dev(d["reg_src"])
    gn = engine.GNTracker(d["st"], d["fs"], _gp(d), d["cfg_reg_lm_lambda"], src.shape[0])
    gn.bricks = ops.BrickCache(d["neighbor_dx"], int(d["num_nei_cells"])).build(d["st"])
    ct, keep = ops.color_term(d["fc"], U.dev(d["reg_colors"]), photometric=True, photo_weight=d["photometric_loss_weight"])
    T, cnt, res_cm, iters, valid, extra = gn.track(src, d["reg_Tinit"], int(d["cfg_reg_iter_n"]),
                                                   term_deg=d["cfg_reg_term_thre_deg"], term_m=d["cfg_reg_term_thre_m"],
                                                   color=ct)
    assert valid == bool(d["trk_valid"])
    np.testing.assert_allclose(T[:3, 3], d["trk_T"][:3, 3], rtol=0, atol=1e-4)
    np.testing.assert_allclose(T[:3, :3], d["trk_T"][:3, :3], rtol=0, atol=1e-5)


def test_color_mapping_two_iterations(cg):
    """Mapper.mapping with colour_on: gradients of geo / colour features and of both decoders for
    two iterations, post-Adam parameters (mapper.py:645-818, loss.py:31-42)."""
    from pin_slam_amd import ops
    from tests import gpu_util as U
    d = cg
    k = int(d["query_nn_k"])
    geo, col = U.dev(d["local_geo_features"]), U.dev(d["local_color_features"])
    gdec, cdec = U.dev(d["dec_flat"]), U.dev(d["cdec_flat"])
    cert = U.dev(d["local_point_certainties"]); tsu = U.dev(d["local_point_ts_update"], torch.int32)
    fs = dataclasses.replace(d["fs"], feats=geo, dec=gdec, certainty=cert)
    fc = dataclasses.replace(d["fc"], feats=col, dec=cdec, certainty=cert)
    g = {n: torch.zeros_like(t) for n, t in (("geo", geo), ("col", col), ("gdec", gdec), ("cdec", cdec))}
    m = {n: torch.zeros_like(t) for n, t in g.items()}
    v = {n: torch.zeros_like(t) for n, t in g.items()}
    params = {"geo": geo, "col": col, "gdec": gdec, "cdec": cdec}
    bs = d["map_coord0"].shape[0]
    buf = ops.TrainBuffers(bs, int(d["map_dec"]), k, 64, 1)
    for it in range(2):
        lab, w = U.dev(d[f"map_label{it}"]), U.dev(d[f"map_w{it}"])
        ops.train_step(d["st"], fs, buf, U.dev(d[f"map_coord{it}"]), lab, w, U.dev(d[f"map_ts{it}"], torch.int32), cert, tsu,
                       g["geo"], g["gdec"], sigma=d["sdf_scale"], weight_e=d["map_weight_e"], eik_eps=d["map_eps"])
        ops.train_color_step(fc, buf, lab, U.dev(d[f"map_color{it}"]), w, g["col"], g["cdec"],
                             surface_range=d["surface_sample_range_m"], weight_i=d["weight_i"])
        for name, key in (("geo", "gfeat"), ("col", "cfeat"), ("gdec", "gdec"), ("cdec", "cdec")):
            ref = d[f"map_{key}{it}"]
            assert np.max(np.abs(g[name].cpu().numpy() - ref)) < 4e-4 * np.abs(ref).max(), (name, it)
        for name in params:
            ops.adam_step(params[name], g[name], m[name], v[name], it + 1, d["map_lr"], eps=d["map_adam_eps"])
    for name, key in (("geo", "map_geo_after"), ("col", "map_color_after"), ("gdec", "map_gdec_after"), ("cdec", "map_cdec_after")):
        diff = np.abs(params[name].cpu().numpy() - d[key])
        assert np.mean(diff < 1e-4) > 0.99, name
