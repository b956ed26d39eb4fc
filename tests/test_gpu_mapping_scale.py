"""WHOLE `Mapper.mapping` calls at scale against the oracle (VERDICT r3 "missing" #2 and #4; BASELINE config 4):

* one GPU, the large-batch path end to end: a 2^17 batch on the c3 bench map (2.23 M rows), two iterations, with the forms the
  2^20 path strings together all active in one call -- pool-record reuse (pin_gather_records_drawn), the row-parallel lazy Adam
  (pin_adam_lazy_prepare_rows), the recomputing weight gradient (train_dw_recompute_kernel: >= 8192 tiles) -- and the END
  STATE (features, decoder, certainty, ts_update) held against the numpy oracle's whole-batch run (float64, dense Adam);
* the spatially sharded mapper (pin_slam_amd.dp) at the same size: 2 and 4 ranks sharing the GPU (HostStagedComm), every rank
  bit-identical to the others, rank 0's end state against (i) the one-rank product run on the same batches and (ii) the oracle;
* the same with the colour branch on the c5 map (5.3 M rows, SDF + colour decoders), 2^16 samples, one GPU and 2 ranks;
* the replica-consistency check: ranks whose replicas differ in one drawn index refuse to exchange rows.

The reference's batches are whatever torch.randint hands it (utils/mapper.py:462-480); here every run is seeded alike, the drawn
indices are recorded by the one-rank run and the oracle is fed exactly those (tests/_map_scale_worker.py)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from oracle import pin_oracle as O

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ITERS = 2
CASES = {"c3": dict(bs=1 << 17, worlds=(2, 4)), "c5": dict(bs=1 << 16, worlds=(2,))}


def _launch(tmp_path, workload, world, bs, extra=()):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    outs = [str(tmp_path / f"{workload}_{world}_{r}{'_' + extra[0] if extra else ''}.npz") for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_map_scale_worker.py"), str(r), str(world), str(port), outs[r],
                               workload, str(bs), str(ITERS), *extra]) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=900) == 0
    return [np.load(o) for o in outs]


def _oracle_call(workload, bs, one):
    """The reference's Mapper.mapping(ITERS) on the recorded batches: oracle train_step (float64) + dense Adam on every row, the
    certainty / ts side effects of the main queries (neural_points.py:685-710); colour branch for colour maps."""
    import bench
    from pin_slam_amd import synth
    from tests._map_scale_worker import POOL
    wl = bench.WORKLOADS[workload]
    cfg = wl["cfg"]
    H, L, k, res = wl["hidden"], wl["levels"], int(cfg["query_nn_k"]), float(cfg["voxel_size_m"])
    colour = bool(wl.get("color", False))
    m = synth.build_map(layers=wl["layers"], resolution=res, **wl["map"])  # (for the pool generator: the sheets' geometry)
    pool_c, pool_l = synth.make_pool(m, n=POOL[workload], sigma=wl.get("pool_sigma", 0.25))
    pool_rgb = np.random.default_rng(9).random((len(pool_l), 3), dtype=np.float32) if colour else None
    positions = one["pos"]  # the map as NeuralPoints.update built it (it drops the points whose hash slot is taken)
    table64 = np.full(int(one["table_size"]), -1, np.int64)
    table64[one["table_slots"]] = one["table_vals"]
    del m
    dx, mv = O.search_neighborhood(2, float(cfg["search_alpha"]), res)
    sdf_scale = 0.55 * float(cfg.get("sigma_sigmoid_m", 0.1))
    weight_e, eps = float(cfg.get("weight_e", 0.5)), np.float32(res * 0.2)
    surface_range = float(cfg.get("surface_sample_range_m", 0.25))
    feats, flat = one["feat0"].astype(np.float64), one["dec0"].astype(np.float64)
    cert, tsu = one["cert0"].copy(), one["tsu0"].copy()
    st = dict(mf=np.zeros_like(feats), vf=np.zeros_like(feats), md=np.zeros_like(flat), vd=np.zeros_like(flat))
    if colour:
        cfeats, cflat = one["cfeat0"].astype(np.float64), one["cdec0"].astype(np.float64)
        st.update(mc=np.zeros_like(cfeats), vc=np.zeros_like(cfeats), mcd=np.zeros_like(cflat), vcd=np.zeros_like(cflat))
    gfs, gds, gcs = [], [], []
    side = {}

    def make_searcher(table_feats, train_first):
        state = {"train": train_first}

        def searcher(points):
            train = state["train"]
            state["train"] = False
            parts = []
            c, t = cert, tsu
            for a in range(0, len(points), 32768):  # chunks bound the [N, Kc] temporaries; the side effects thread through
                p = points[a:a + 32768]
                s = O.radius_search(p, table64, positions, res, dx, mv)
                qf = O.query_feature(p, s, table_feats.astype(np.float32), positions, c, k, weighted_first=False, training_mode=train,
                                     query_ts=np.zeros(len(p), np.int32) if train else None, ts_update=t)
                if train:
                    c, t = qf["certainties_after"], qf["ts_update_after"]
                parts.append(qf)
            if train:
                side["cert"], side["tsu"] = c, t
            return {key: np.concatenate([q[key] for q in parts]) for key in ("geo_feat", "knn_idx", "knn_d2", "nn_count")}
        return searcher

    for it in range(ITERS):
        idx = one["hist"][it]
        coord, label, w = pool_c[idx], pool_l[idx], np.ones(len(idx), np.float32)
        r = O.train_step(coord, label, w, make_searcher(feats, True), feats, positions, flat, (11, H, L), sdf_scale, k,
                         weighted_first=bool(cfg.get("weighted_first", True)), dec=10, eps=eps, weight_e=weight_e)
        cert, tsu = side["cert"], side["tsu"]
        gfs.append(r["feat_grad"]); gds.append(r["dec_grad"])
        if colour:
            rc = O.train_color_step(coord, label, pool_rgb[idx], w, make_searcher(cfeats, False), cfeats, cflat, (11, H, L, 3), k,
                                    surface_range=surface_range, weight_i=1.0)
            gcs.append(rc["feat_grad"])
            cfeats, st["mc"], st["vc"] = O.adam_step(cfeats, rc["feat_grad"], st["mc"], st["vc"], it + 1)
            cflat, st["mcd"], st["vcd"] = O.adam_step(cflat, rc["dec_grad"], st["mcd"], st["vcd"], it + 1)
        feats, st["mf"], st["vf"] = O.adam_step(feats, r["feat_grad"], st["mf"], st["vf"], it + 1)
        flat, st["md"], st["vd"] = O.adam_step(flat, r["dec_grad"], st["md"], st["vd"], it + 1)
    out = dict(feats=feats, dec=flat, cert=cert, tsu=tsu, gfs=gfs, gds=gds)
    if colour:
        out.update(cfeats=cfeats, cdec=cflat, gcs=gcs)
    return out


STATS = []  # what every comparison measured -> profiles/r04_scale_parity.json when PIN_WRITE_PROFILES is set


def _trained_like(got, ref, ref_grads, start, what, lr=0.01, clean_rel=1e-3, tol=1e-4, min_share=0.99):
    """End state of a table after ITERS Adam steps against the oracle's.

    eps = 1e-15 makes Adam step by ~lr whatever the size of a gradient (step 1 is lr * sign(g) exactly), so an entry whose
    gradient sits at the rounding noise of its accumulation may step the other way (golden_util.adam_outliers).  At this
    scale that has a second-order effect the 512-sample fixtures do not show: a flipped entry moves its feature by 2 lr = 0.02
    (a fifth of the feature spread), which changes ReLU patterns of the few samples that read the row and with them the
    SECOND iteration's gradient of neighbouring, perfectly well-conditioned entries (a row is read by a handful of samples
    per iteration).  So:
      * entries the batches never touch must not have moved at all (bit-exact);
      * entries with a clean gradient in iteration 1 that no later iteration touches test the optimiser arithmetic alone
        (the step, then the gradient-free steps the lazy form replays): they must agree to 1e-5;
      * entries clean in EVERY iteration: at least `min_share` of them within `tol`, all of them within Adam's reach."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    gs = [np.asarray(g).reshape(ref.shape) for g in ref_grads]
    clean = np.ones(ref.shape, bool)
    never = np.ones(ref.shape, bool)
    for g in gs:
        clean &= np.abs(g) >= clean_rel * np.abs(g).max()
        never &= g == 0
    first_only = np.abs(gs[0]) >= clean_rel * np.abs(gs[0]).max()
    for g in gs[1:]:
        first_only &= g == 0
    diff = np.abs(got - ref)
    share = float((diff[clean] < tol).mean())
    stats = dict(what=what, entries=int(ref.size), touched_fraction=float(1 - never.mean()), clean_fraction=float(clean.mean()),
                 clean_within_tol=share, tol=tol, clean_median=float(np.median(diff[clean])), clean_p999=float(np.quantile(diff[clean], 0.999)),
                 worst_clean=float(diff[clean].max()), worst_any=float(diff.max()),
                 first_only_entries=int(first_only.sum()), first_only_worst=float(diff[first_only].max()) if first_only.any() else None)
    print(stats)
    STATS.append(stats)
    assert clean.sum() > 1000 or ref.size < 20000, stats
    assert np.array_equal(got[never], np.asarray(start, np.float64)[never]), f"{what}: an entry the batch never touches moved"
    if first_only.any():
        assert stats["first_only_worst"] < 1e-5, stats
    assert share >= min_share, stats
    assert diff.max() <= 2.0 * lr * len(ref_grads) * 1.05, stats
    return stats


def _check_against_oracle(run, ora, one, colour, tag):
    _trained_like(run["feats"], ora["feats"], ora["gfs"], one["feat0"], f"{tag}: features")
    _trained_like(run["dec"], ora["dec"], ora["gds"], one["dec0"], f"{tag}: decoder", clean_rel=1e-2, tol=2e-4, min_share=0.97)
    np.testing.assert_allclose(run["cert"], ora["cert"], rtol=1e-4, atol=1e-5)
    assert np.array_equal(run["tsu"], ora["tsu"])
    # assign_local_to_global ran: the global tables carry the trained rows (the whole synthetic map is local)
    assert np.array_equal(run["gfeats"], run["feats"]) and np.array_equal(run["gcert"], run["cert"])
    if colour:
        _trained_like(run["cfeats"], ora["cfeats"], ora["gcs"], one["cfeat0"], f"{tag}: colour features")


@pytest.fixture(scope="module", params=list(CASES))
def scale_case(request, tmp_path_factory):
    workload = request.param
    bs = CASES[workload]["bs"]
    tmp = tmp_path_factory.mktemp(f"mapscale_{workload}")
    (one,) = _launch(tmp, workload, 1, bs)
    one = dict(one)
    ora = _oracle_call(workload, bs, one)
    import bench
    return dict(workload=workload, bs=bs, tmp=tmp, one=one, ora=ora, colour=bool(bench.WORKLOADS[workload].get("color", False)))


def test_large_batch_mapping_call_end_state_vs_oracle(scale_case):
    """One GPU: Mapper.mapping(2) at 2^17 (c3) / 2^16 (c5, colour) samples -- record reuse, rows-form lazy Adam and (c3) the
    recomputing weight gradient in ONE call -- end state against the oracle's whole-batch run."""
    c, one = scale_case, scale_case["one"]
    assert bool(one["records_reused"])
    if c["workload"] == "c3":  # 1.7 M records over 2.2 M rows: the row-parallel lazy Adam (c5: 0.6 M over 5.3 M rows -- per record)
        assert int(one["lazy_rows_launches"]) == ITERS
    n_eik = (c["bs"] + 9) // 10
    tiles = (n_eik + 1) // 2 + max(0, c["bs"] - 4 * ((n_eik + 1) // 2) + 15) // 16  # train_fused.h fused_tiles()
    if c["workload"] == "c3":
        assert tiles >= 8192  # train.hip DW_RECOMPUTE_MIN_TILES: the weight gradient recomputes the layers' inputs
    _check_against_oracle(one, c["ora"], one, c["colour"], f"{c['workload']} one GPU")


def test_spatial_shards_at_scale(scale_case):
    """2 (and 4) ranks sharing the GPU, spatial shards, host-staged exchange, the same recorded batches: ranks bit-identical to
    each other; rank 0 against the oracle and against the one-rank product run."""
    c, one, ora = scale_case, scale_case["one"], scale_case["ora"]
    for world in CASES[c["workload"]]["worlds"]:
        rs = _launch(c["tmp"], c["workload"], world, c["bs"])
        r0 = rs[0]
        assert np.array_equal(r0["hist"], one["hist"])  # the same batches as the one-rank run and the oracle
        assert all(bool(r["records_reused"]) for r in rs)  # every rank searched the pool samples of its box once
        keys = [k for k in r0.files if k.startswith("sha_")]
        assert len(keys) >= 6
        for r in rs[1:]:
            for key in keys:
                assert str(r[key]) == str(r0[key]), f"world {world}: {key} differs between ranks"
        n_eik = (c["bs"] + 9) // 10
        for it in range(ITERS):  # the boxes cut every batch: nothing lost, nothing doubled
            assert sum(int(r["n_main"][it]) for r in rs) == c["bs"] and sum(int(r["n_eik"][it]) for r in rs) == n_eik
        assert 0 < int(r0["n_halo"]) < 0.2 * len(one["feat0"])
        _check_against_oracle(r0, ora, one, c["colour"], f"{c['workload']} {world} ranks")
        # (ii) against the one-rank product run, where the gradients are well above the noise in every iteration
        clean = np.ones(one["feats"].shape, bool)
        for g in ora["gfs"]:
            clean &= np.abs(g) >= 1e-3 * np.abs(g).max()
        d1 = np.abs(r0["feats"].astype(np.float64) - one["feats"])[clean]
        STATS.append(dict(what=f"{c['workload']} {world} ranks vs one rank: features", clean_within_tol=float((d1 < 1e-4).mean()),
                          clean_median=float(np.median(d1)), worst_clean=float(d1.max())))
        print(STATS[-1])
        assert (d1 < 1e-4).mean() >= 0.99
        np.testing.assert_allclose(r0["cert"], one["cert"], rtol=1e-4, atol=1e-5)


def test_dense_shards_at_scale(scale_case):
    """2 ranks sharing the GPU, DENSE shards (north_star's literal split: contiguous halves of every drawn batch, one whole-table
    all-reduce of [decoder | features] per iteration, replicated dense Adam) through the drop-in Mapper on the same recorded
    batches -- on the colour map (c5) with the colour table's own exchange and the whole batch's surface-sample count: ranks
    bit-identical, rank 0 against the oracle like the spatial shards."""
    c, one, ora = scale_case, scale_case["one"], scale_case["ora"]
    rs = _launch(c["tmp"], c["workload"], 2, c["bs"], extra=("dense",))
    r0 = rs[0]
    assert np.array_equal(r0["hist"], one["hist"])
    keys = [k for k in r0.files if k.startswith("sha_")]
    assert len(keys) >= 6
    for key in keys:
        assert str(rs[1][key]) == str(r0[key]), f"{key} differs between the ranks"
    _check_against_oracle(r0, ora, one, c["colour"], f"{c['workload']} 2 ranks, dense shards")


def test_diverged_replicas_refuse_to_exchange(tmp_path):
    """only the boxes are agreed by an exchange; every other list of the spatial mapper is derived per rank from its
    replica.  One changed index on rank 1 -> the signature exchange (pin_dp_signature) stops BOTH ranks with a diagnostic."""
    rs = _launch(tmp_path, "c3", 2, 1 << 14, extra=("diverge",))
    for r in rs:
        assert "replicas" in str(r["err"]) and "first-batch checksum" in str(r["err"]), str(r["err"])


def test_write_scale_parity_record():
    """(last in the file) the numbers the comparisons above measured, for profiles/r04_scale_parity.json."""
    import json
    out = os.environ.get("PIN_SCALE_PARITY_OUT")
    if out and STATS:
        with open(out, "w") as f:
            json.dump(STATS, f, indent=1)
