"""Parity of the BENCHMARKED kernels, and parity at BASELINE scale:

* the four-lanes-per-query GN tile kernel (gn_accumulate_quad_kernel, split-fp16 default and PIN_MLP=f32) against
  the REFERENCE's per-point SDF / gradient / validity mask on the fixtures (it was only checked through its sums);
* the HIP path against the numpy oracle on the bench workload itself: 2.2 M neural points, a 5e7-slot table,
  Kc = 81, k = 8, decoder 4x64 -- neighbour indices bit-exact, SDF / gradient 1e-4, one Gauss-Newton step,
  one training step (gradients and scalar losses)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import pin_oracle as O
from tests import golden_util as G

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _check_points(d, sdf, grad, nn):
    m = d["qsp_mask"]
    assert np.array_equal(nn >= d["track_mask_query_nn_k"], m)
    assert m.sum() > 100
    np.testing.assert_allclose(sdf[m], d["qsp_sdf"][m], rtol=1e-4, atol=2e-6)
    scale = np.abs(d["qsp_grad"]).max(1, keepdims=True) + 1e-6
    assert np.max((np.abs(grad - d["qsp_grad"]) / scale)[m]) < 1e-4


@pytest.mark.parametrize("case", ["c2_wf", "c3_bigtable"])
@pytest.mark.parametrize("mlp", ["h2", "f32"])
def test_gn_tile_kernel_points_vs_reference(tmp_path, case, mlp):
    """pin_gn_accumulate(want_points) -> gn_accumulate_quad_kernel: per-point outputs vs Tracker.query_source_points
    of the reference (tracker.py:297-354).  The decoder arithmetic is an environment choice read once per process,
    so each variant runs in a worker process."""
    out = str(tmp_path / "pts.npz")
    env = dict(os.environ, PIN_MLP=mlp)
    subprocess.run([sys.executable, os.path.join(HERE, "_variant_worker.py"), out, "fixture", case], check=True, env=env,
                   timeout=600)
    r = np.load(out)
    d = G.load(case)
    _check_points(d, r["sdf"], r["grad"], r["nn"])
    # and the sums of the same launch reproduce the per-point reduction of the oracle on the reference's values
    reg = O.registration_step(d["query"], d["qsp_sdf"], d["qsp_grad"], d["qsp_std"], r["nn"],
                              valid_nn_k=int(d["track_mask_query_nn_k"]), min_grad_norm=d["cfg_reg_min_grad_norm"],
                              max_grad_norm=d["cfg_reg_max_grad_norm"],
                              max_sdf_std=d["cfg_surface_sample_range_m"] * d["cfg_max_sdf_std_ratio"],
                              GM_dist=d["cfg_reg_GM_dist_m"], GM_grad=d["cfg_reg_GM_grad"], lm_lambda=d["cfg_reg_lm_lambda"])
    assert abs(int(round(r["sums"][29])) - reg["valid_count"]) <= 2


# --------------------------------------------------------------------------------------------- BASELINE scale
@pytest.fixture(scope="module")
def big():
    from pin_slam_amd import ops, synth
    H, L, k = 64, 4, 8
    m = synth.build_map(layers=16)  # ~2.2 M neural points, 5e7 slots (bench.py workload c3)
    P = len(m.positions)
    assert P > 2_000_000
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    pos = dev(m.positions)
    pos4 = torch.empty((P, 4), dtype=torch.float32, device="cuda")
    ops.pack_positions(pos, torch.zeros(P, dtype=torch.int32, device="cuda"), pos4)
    dx, mv = ops.search_neighborhood(2, 0.5, 0.4)
    g2l = torch.arange(P + 1, dtype=torch.int32, device="cuda")
    g2l[-1] = -1
    st = ops.SearchState(table=dev(m.table), pos4=pos4, cand_off=dev(ops.candidate_offsets(dx, m.buffer_size)), n_points=P,
                         resolution=0.4, max_valid_dist2=mv, travel_dist=torch.zeros(1, device="cuda"), cur_ts=0,
                         diff_travel_dist_local=410.0, global2local=g2l)
    decf = synth.init_decoder(H, L)
    fs = ops.FieldState(feats=dev(m.features), dec=dev(decf), k=k, hidden=H, levels=L, weighted_first=True,
                        sdf_scale=0.055, certainty=torch.zeros(P, device="cuda"), pos=pos)
    bricks = ops.BrickCache(dx, 2).build(st, wait=True)
    odx, omv = O.search_neighborhood(2, 0.5, 0.4)
    assert np.array_equal(odx, dx)
    return dict(m=m, st=st, fs=fs, bricks=bricks, dx=odx, mv=omv, table64=m.table.astype(np.int64), H=H, L=L, k=k,
                params=O.unpack_decoder(decf, 11, H, L), decf=decf, P=P)


def _oracle_search(b, pts):
    return O.radius_search(pts, b["table64"], b["m"].positions, 0.4, b["dx"], b["mv"])


@pytest.mark.parametrize("use_bricks", [True, False])
def test_scale_knn_and_sdf_vs_oracle(big, use_bricks):
    from pin_slam_amd import ops, synth
    from tests import gpu_util as U
    b = big
    q = synth.make_scan(b["m"], n=4096, seed=11)
    nbr, nn, _ = ops.knn_query(b["st"], U.dev(q), b["k"], bricks=b["bricks"] if use_bricks else None)
    vec, idx, _ = U.nbr_split(nbr)
    s = _oracle_search(b, q)
    qf = O.query_feature(q, s, b["m"].features, b["m"].positions, None, b["k"])
    assert np.array_equal(idx, qf["knn_idx"].astype(np.int32)), "neighbour indices differ from the oracle at 2.2 M points"
    assert np.array_equal(nn.cpu().numpy(), qf["nn_count"])
    assert (qf["nn_count"] >= 6).mean() > 0.9
    sdf, grad, _, _ = ops.sdf_query(b["fs"], U.dev(q), nbr, nn)
    rs, rg, _, _, _ = O.query_sdf(q, s, b["m"].features, b["m"].positions, b["params"], 0.055, b["k"])
    has = qf["nn_count"] > 0
    np.testing.assert_allclose(sdf.cpu().numpy()[has], rs[has], rtol=1e-4, atol=2e-6)
    scale = np.abs(rg).max(1, keepdims=True) + 1e-6
    assert np.max((np.abs(grad.cpu().numpy() - rg) / scale)[has]) < 1e-4


def test_scale_gn_step_vs_oracle(big):
    """One registration step over 4096 scan points on the bench map: the tile kernel's per-point outputs and its
    Gauss-Newton increment against the oracle's."""
    from pin_slam_amd import ops, synth
    from pin_slam_amd._lib import GnParams
    from tests import gpu_util as U
    b = big
    q = synth.make_scan(b["m"], n=4096, seed=12)
    T0 = np.eye(4)
    T0[:3, 3] = (0.03, -0.02, 0.01)
    nbr, nn, cur = ops.knn_query(b["st"], U.dev(q), b["k"], pose=T0, bricks=b["bricks"])
    gp = GnParams()  # random-init decoder: tiny gradients, so the norm window is opened up (as tests/_variant_worker.py)
    gp.valid_nn_k, gp.min_grad_norm, gp.max_grad_norm, gp.max_sdf_std, gp.gm_dist, gp.gm_grad = 6, 1e-5, 1e3, 0.25, 0.3, 0.1
    sums, sdf, grad = ops.gn_accumulate(b["fs"], gp, cur, nbr, nn, want_points=True)
    curh = cur.cpu().numpy()
    s = _oracle_search(b, curh)
    rs, rg, rstd, rnn, _ = O.query_sdf(curh, s, b["m"].features, b["m"].positions, b["params"], 0.055, b["k"])
    assert np.array_equal(nn.cpu().numpy(), rnn)
    has = rnn >= 6
    np.testing.assert_allclose(sdf.cpu().numpy()[has], rs[has], rtol=1e-4, atol=2e-6)
    scale = np.abs(rg).max(1, keepdims=True) + 1e-6
    assert np.max((np.abs(grad.cpu().numpy() - rg) / scale)[has]) < 1e-4
    reg = O.registration_step(curh, rs, rg, rstd, rnn, valid_nn_k=6, min_grad_norm=1e-5, max_grad_norm=1e3, max_sdf_std=0.25,
                              GM_dist=0.3, GM_grad=0.1, lm_lambda=1e-4)
    T, cnt, res_cm, _ = ops.solve_gn(sums.cpu().numpy(), 1e-4)
    assert abs(cnt - reg["valid_count"]) <= 2 and cnt > 3000
    np.testing.assert_allclose(T, reg["T"], rtol=0, atol=1e-5)
    assert abs(res_cm - reg["residual_cm"]) < 1e-4 * max(1.0, reg["residual_cm"])


def test_scale_training_step_vs_oracle(big):
    """One Mapper.mapping iteration of 2048 samples (+ 6 x 205 Eikonal probes) on the bench map: feature / decoder
    gradients and the two loss terms against the oracle (float64)."""
    from pin_slam_amd import ops, synth
    from tests import gpu_util as U
    b = big
    bs, dec = 2048, 10
    coord, label = synth.make_pool(b["m"], n=bs, seed=7)
    feats = b["fs"].feats
    gfeat, gdec = torch.zeros_like(feats), torch.zeros_like(b["fs"].dec)
    cert = torch.zeros(b["P"], device="cuda")
    tsu = torch.zeros(b["P"], dtype=torch.int32, device="cuda")
    import dataclasses
    fs = dataclasses.replace(b["fs"], certainty=cert)
    buf = ops.TrainBuffers(bs, dec, b["k"], b["H"], b["L"])
    loss = ops.train_step(b["st"], fs, buf, U.dev(coord), U.dev(label), torch.ones(bs, device="cuda"),
                          torch.zeros(bs, dtype=torch.int32, device="cuda"), cert, tsu, gfeat, gdec, sigma=0.055,
                          weight_e=0.5, eik_eps=0.08, bricks=b["bricks"])

    def searcher(p):
        return O.query_feature(p, _oracle_search(b, p), b["m"].features, b["m"].positions, None, b["k"], weighted_first=False)

    r = O.train_step(coord, label, np.ones(bs, np.float32), searcher, b["m"].features, b["m"].positions, b["decf"],
                     (11, b["H"], b["L"]), 0.055, b["k"], dec=dec, eps=0.08, weight_e=0.5)
    gf, gd = gfeat.cpu().numpy(), gdec.cpu().numpy()
    assert np.max(np.abs(gf - r["feat_grad"])) < 1e-4 * np.abs(r["feat_grad"]).max()
    assert np.max(np.abs(gd - r["dec_grad"])) < 1e-4 * np.abs(r["dec_grad"]).max()
    touched = np.abs(r["feat_grad"]).max(1) > 0
    assert np.array_equal(np.abs(gf).max(1) > 0, touched) and 10_000 < touched.sum() < 30_000
    l_bce, l_eik = loss.cpu().numpy()
    assert abs(l_bce / bs - r["sdf_loss"]) < 1e-5 * abs(r["sdf_loss"])
    assert abs(l_eik / buf.n_eik - r["eik_loss"]) < 1e-4 * abs(r["eik_loss"])
