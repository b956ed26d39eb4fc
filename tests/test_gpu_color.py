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


@pytest.mark.parametrize("form,overlap,frozen", [("two streams", None, False), ("in line", False, False), ("frozen decoders", None, True)])
def test_grouped_colour_iterations_behind_the_abi(cg, form, overlap, frozen):
    """engine.MapTrainer.step_group with a colour decoder: the iterations of a group queued by ONE foreign call
    (pin_train_group_steps) train like the same iterations stepped one by one from Python -- same launches, same operands; what
    differs is the order of the float atomics.  Two streams: the SDF term's weight gradient and decoder step on a side stream beside
    the colour term (the default of a colour map); in line: everything on the caller's stream; frozen decoders (every frame of a run
    after freeze_after_frame, utils/tools.py:263-292): no rider, no weight gradient -- the decoders must not move."""
    from pin_slam_amd import _lib, engine, ops
    from tests import gpu_util as U
    d = cg
    bs, iters = d["map_coord0"].shape[0], 4
    batches = [0, 1, 0, 1]
    results = []
    for arm in ("python", "group"):
        geo, col = U.dev(d["local_geo_features"]), U.dev(d["local_color_features"])
        cert, tsu = U.dev(d["local_point_certainties"]), U.dev(d["local_point_ts_update"], torch.int32)
        fs = dataclasses.replace(d["fs"], feats=geo, dec=U.dev(d["dec_flat"]), certainty=cert, dec_image=None)
        fc = dataclasses.replace(d["fc"], feats=col, dec=U.dev(d["cdec_flat"]), certainty=cert, dec_image=None)
        t = engine.MapTrainer(d["st"], fs, None, None, None, None, tsu, bs=bs, decimation=int(d["map_dec"]), sigma=d["sdf_scale"],
                              weight_e=d["map_weight_e"], eik_eps=d["map_eps"], lr=d["map_lr"], adam_eps=d["map_adam_eps"],
                              loss_weight_on=bool(d.get("map_loss_weight_on", False)), train_decoder=not frozen)
        t.overlap_weight_grad = overlap
        t.set_color(fc, surface_range=d["surface_sample_range_m"], weight_i=d["weight_i"], train_decoder=not frozen)
        assert t.buf.group >= iters
        t.reset_optimizer(iters)
        t.begin_side_effects()
        assert t._two_streams() == (form == "two streams")
        coords = [U.dev(d[f"map_coord{b}"]) for b in batches]
        for j, c in enumerate(coords):  # the group's queries, slot by slot (Mapper.mapping's gather launch writes them)
            t.buf.select(j)
            ops.check(_lib.lib().pin_train_make_queries(ops._ptr(c), t.buf.n_main, t.buf.n_eik, t.buf.dec, t.buf.eik_first,
                                                        float(np.float32(d["map_eps"])), ops._ptr(t.buf.query), ops._stream()), "make_queries")
        t.knn_group(iters)
        lab = torch.stack([U.dev(d[f"map_label{b}"]) for b in batches]).contiguous()
        w = torch.stack([U.dev(d[f"map_w{b}"]) for b in batches]).contiguous()
        ts = torch.stack([U.dev(d[f"map_ts{b}"], torch.int32) for b in batches]).contiguous()
        colr = torch.stack([U.dev(d[f"map_color{b}"]) for b in batches]).contiguous()
        if arm == "group":
            assert t.can_step_group(colr)
            t.step_group(lab, w, ts, 1, iters, color_label=colr)
            assert t._wg_pending == (form == "two streams") and t.lazy.t == iters and t.lazy_c.t == iters
        else:
            for j in range(iters):
                t.buf.select(j)
                t.step_batch(coords[j], lab[j], w[j], ts[j], j + 1, color_label=colr[j], queries_ready=True, knn_ready=True)
        t.buf.select(0)
        t.finish_optimizer()
        torch.cuda.synchronize()
        results.append([x.clone() for x in (geo, col, fs.dec, fc.dec, cert)])
    for name, a, b in zip(("features", "colour features", "decoder", "colour decoder", "certainty"), *results):
        assert not torch.equal(a, torch.zeros_like(a))
        if "decoder" in name:
            assert (a - b).abs().max().item() < 1e-3, name
            if frozen:
                assert torch.equal(a, U.dev(d["dec_flat" if name == "decoder" else "cdec_flat"])) and torch.equal(a, b), name
        elif name == "certainty":
            torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-4)
        else:  # (Adam with eps = 1e-15 turns the rounding noise of a near-zero gradient into a step of ~lr: bound their share)
            assert (a - b).abs().mean().item() < 1e-5 and ((a - b).abs() > 5e-3).float().mean().item() < 1e-3, name
            assert not torch.equal(a, U.dev(d["local_geo_features" if name == "features" else "local_color_features"])), name
