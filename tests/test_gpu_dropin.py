"""GPU tests of the drop-in classes (stand-alone PinConfig, no reference tree needed):
map maintenance kernels (K8-K10) bit-exact against the reference's recorded arrays, and a
small end-to-end loop update -> mapping -> tracking through the reference's call surface."""
import os

import numpy as np
import pytest
import torch

from oracle import pin_oracle as O
from tests import golden_util as G

pytestmark = pytest.mark.gpu


def _cfg(**kw):
    from pin_slam_amd.config import PinConfig
    base = dict(voxel_size_m=0.4, buffer_size=int(5e7), local_map_radius=20.0, local_map_travel_dist_ratio=1.0)
    base.update(kw)
    return PinConfig(**base)


def test_voxel_downsample_matches_reference():
    from pin_slam_amd import _lib
    d = G.load("update")
    L = _lib.lib()
    for ts in range(4):
        pts = torch.from_numpy(d[f"pts{ts}"]).cuda()
        n = pts.shape[0]
        ws = torch.empty(L.pin_maint_workspace_bytes(n), dtype=torch.uint8, device="cuda")
        sel = torch.empty(n, dtype=torch.int32, device="cuda")
        cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
        _lib.check(L.pin_voxel_downsample(pts.data_ptr(), n, float(np.float32(d["resolution"])), sel.data_ptr(),
                                          cnt.data_ptr(), ws.data_ptr(), ws.numel(), None), "vds")
        c = int(cnt.item())
        assert c == len(d[f"sel{ts}"])
        assert np.array_equal(sel[:c].cpu().numpy(), d[f"sel{ts}"].astype(np.int32))


def test_voxel_downsample_fast_form():
    """pin_voxel_downsample_fast (voxel id and tie-breaking value in one sort key, statistics from per-block partials):
    the reference's selection on the fixture frames and on a large random cloud (against the general form); a cloud whose
    voxel ids do not fit the key is REPORTED (count -1) and the wrapper falls back to the general form."""
    from pin_slam_amd import _lib, preprocess
    d = G.load("update")
    L = _lib.lib()

    def run(fn, pts, vs):
        n = pts.shape[0]
        ws = torch.empty(L.pin_maint_workspace_bytes(n), dtype=torch.uint8, device="cuda")
        sel = torch.full((n,), -7, dtype=torch.int32, device="cuda")
        cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
        _lib.check(fn(pts.data_ptr(), n, float(np.float32(vs)), sel.data_ptr(), cnt.data_ptr(), ws.data_ptr(), ws.numel(), None), "vds")
        c = int(cnt.item())
        return c, (sel[:c].cpu().numpy() if c >= 0 else None)

    for ts in range(4):
        pts = torch.from_numpy(d[f"pts{ts}"]).cuda()
        c, sel = run(L.pin_voxel_downsample_fast, pts, d["resolution"])
        assert c == len(d[f"sel{ts}"]) and np.array_equal(sel, d[f"sel{ts}"].astype(np.int32))
    g = torch.Generator(device="cuda").manual_seed(0)
    for n, vs, spread in ((300_001, 0.08, 60.0), (1_000_000, 0.4, 150.0), (17, 0.5, 3.0), (1, 0.5, 1.0)):
        pts = ((torch.rand((n, 3), device="cuda", generator=g) - 0.5) * spread).contiguous()
        pts[n // 2:] = pts[n // 2:] * 0.05  # a dense core: many points per voxel, ties on the quantised distance
        a, b = run(L.pin_voxel_downsample_fast, pts, vs), run(L.pin_voxel_downsample, pts, vs)
        assert a[0] == b[0] > 0 and np.array_equal(a[1], b[1]), (n, vs)
    wide = ((torch.rand((50_000, 3), device="cuda", generator=g) - 0.5) * 4000.0).contiguous()  # 2^17 voxels per axis at 3 cm
    assert run(L.pin_voxel_downsample_fast, wide, 0.03)[0] == -1
    ref = run(L.pin_voxel_downsample, wide, 0.03)
    idx = preprocess._voxel_down_sample_i32(wide, 0.03)
    assert np.array_equal(idx.cpu().numpy(), ref[1])


def test_update_reset_local_map_bit_exact():
    """NeuralPoints.update / reset_local_map over 4 frames: point count, positions, timestamps,
    hash table, local mask and global2local equal the reference's (neural_points.py:311-513)."""
    from pin_slam_amd.dropin.model.neural_points import NeuralPoints
    d = G.load("update")
    npts = NeuralPoints(_cfg())
    npts.travel_dist = torch.from_numpy(d["travel_dist"]).cuda()
    for ts in range(4):
        npts.update(torch.from_numpy(d[f"pts{ts}"]).cuda(), torch.tensor([9.0 * ts, 0.0, 0.0]), torch.eye(3), ts)
        assert npts.count() == d[f"count{ts}"]
        assert np.array_equal(npts.local_mask.cpu().numpy(), d[f"local_mask{ts}"])
        assert np.array_equal(npts.global2local.cpu().numpy(), d[f"global2local{ts}"])
        M = int(d[f"local_mask{ts}"][:-1].sum())
        assert npts.local_count() == M and npts.local_geo_features.shape == (M + 1, 8)
        lp = npts.neural_points[npts.local_mask[:-1]]
        assert torch.equal(lp, npts.local_neural_points)
    assert np.array_equal(npts.neural_points.cpu().numpy(), d["neural_points"])
    assert np.array_equal(npts.point_ts_create.cpu().numpy(), d["point_ts_create"])
    tab = npts.buffer_pt_index.cpu().numpy()
    slots = np.nonzero(tab >= 0)[0]
    assert np.array_equal(slots, d["table_slots"])
    assert np.array_equal(tab[slots], d["table_vals"].astype(np.int32))


def test_assign_local_to_global_and_query_api():
    from pin_slam_amd.dropin.model.decoder import Decoder
    from pin_slam_amd.dropin.model.neural_points import NeuralPoints
    d = G.load("update")
    cfg = _cfg(search_alpha=0.5, query_nn_k=8, feature_std=0.1)
    npts = NeuralPoints(cfg)
    npts.travel_dist = torch.from_numpy(d["travel_dist"]).cuda()
    for ts in range(2):
        npts.update(torch.from_numpy(d[f"pts{ts}"]).cuda(), torch.tensor([9.0 * ts, 0.0, 0.0]), torch.eye(3), ts)
    with torch.no_grad():
        npts.local_geo_features.data.add_(1.0)
    npts.local_point_certainties.add_(2.0)
    mask = npts.local_mask
    before = npts.geo_features.clone()
    npts.assign_local_to_global()
    after = npts.geo_features
    assert torch.allclose(after[mask], before[mask] + 1.0)
    assert torch.equal(after[~mask], before[~mask])
    assert torch.allclose(npts.point_certainties[mask[:-1]], torch.full((int(mask[:-1].sum()),), 2.0, device="cuda"))
    # tensor API (Mesher-style): query_feature -> Decoder.sdf, against the oracle
    dec = Decoder(cfg, 32, 2, 1)
    q = torch.from_numpy(d["pts1"][:500] + 0.03).cuda()
    feat, _, w, nn, cert = npts.query_feature(q, training_mode=False, query_locally=True)
    sdf = dec.sdf(feat)
    table = npts.buffer_pt_index.cpu().numpy().astype(np.int64)
    dx, mv = O.search_neighborhood(2, 0.5, 0.4)
    s = O.radius_search(q.cpu().numpy(), table, npts.neural_points.cpu().numpy(), 0.4, dx, mv,
                        ts_create=npts.point_ts_create.cpu().numpy(), travel_dist=d["travel_dist"], cur_ts=1,
                        diff_travel_dist_local=npts.diff_travel_dist_local)
    qf = O.query_feature(q.cpu().numpy(), s, npts.local_geo_features.data.cpu().numpy(),
                         npts.local_neural_points.cpu().numpy(), npts.local_point_certainties.cpu().numpy(), 8,
                         global2local=npts.global2local.cpu().numpy())
    assert np.array_equal(nn.cpu().numpy(), qf["nn_count"])
    np.testing.assert_allclose(feat.cpu().numpy(), qf["geo_feat"], rtol=1e-5, atol=3e-7)
    params = O.unpack_decoder(dec.flat_params().cpu().numpy(), 11, 32, 2)
    ref = dec.sdf_scale * O.mlp_forward(qf["geo_feat"].astype(np.float64),
                                        tuple([x.astype(np.float64) for x in p] if isinstance(p, list) else p.astype(np.float64) for p in params))[:, 0]
    np.testing.assert_allclose(sdf.cpu().numpy(), ref, rtol=1e-4, atol=1e-6)
    assert list(dec.state_dict().keys()) == ["layers.0.weight", "layers.0.bias", "layers.1.weight", "layers.1.bias",
                                             "lout.weight", "lout.bias"]


class _FakeDataset:
    lose_track = False
    stop_status = False


def test_mini_slam_loop_update_map_track():
    """update -> Mapper.mapping -> Tracker.tracking through the reference's call surface on a
    synthetic sheet: training lowers the SDF error, registration recovers a known offset."""
    from pin_slam_amd import synth
    from pin_slam_amd.dropin.model.decoder import Decoder
    from pin_slam_amd.dropin.model.neural_points import NeuralPoints
    from pin_slam_amd.dropin.utils.mapper import Mapper
    from pin_slam_amd.dropin.utils.tracker import Tracker
    torch.manual_seed(0)
    cfg = _cfg(search_alpha=0.5, query_nn_k=8, bs=4096, local_map_radius=40.0, local_map_travel_dist_ratio=5.0,
               reg_iter_n=30)
    rng = np.random.default_rng(0)
    pts, _ = synth.disc_points(rng, 120_000, 25.0, 2)
    npts = NeuralPoints(cfg)
    npts.travel_dist = torch.zeros(1, device="cuda")
    npts.update(torch.from_numpy(pts).cuda(), torch.zeros(3), torch.eye(3), 0)
    assert npts.count() > 10_000 and npts.local_count() == npts.count()
    dec = Decoder(cfg, 64, 1, 1)
    decoders = {"sdf": dec, "semantic": None, "color": None}
    mp = Mapper(cfg, _FakeDataset(), npts, decoders)
    base, _ = synth.disc_points(rng, 400_000, 24.0, 2)
    nrm = synth.sheet_normal(base[:, 0].astype(np.float64), base[:, 1].astype(np.float64))
    dd = 0.15 * rng.standard_normal(len(base))
    mp.global_coord_pool = torch.from_numpy((base + dd[:, None] * nrm).astype(np.float32)).cuda()
    mp.coord_pool = mp.global_coord_pool
    mp.sdf_label_pool = torch.from_numpy(dd.astype(np.float32)).cuda()
    mp.weight_pool = torch.ones(len(base), device="cuda")
    mp.time_pool = torch.zeros(len(base), dtype=torch.int, device="cuda")
    mp.pool_sample_count = len(base)
    probe = mp.global_coord_pool[:20000]
    lab = mp.sdf_label_pool[:20000]
    err0 = (mp.sdf(probe)[0] - lab).abs().mean().item()
    mp.mapping(300)
    err1 = (mp.sdf(probe)[0] - lab).abs().mean().item()
    assert err1 < 0.35 * err0 and err1 < 0.05, (err0, err1)
    assert torch.allclose(npts.geo_features[:-1], npts.local_geo_features.data[:-1])  # assign_local_to_global ran
    trk = Tracker(cfg, npts, decoders)
    scan, _ = synth.disc_points(rng, 20_000, 20.0, 2)
    T_true = np.eye(4)
    a = 0.004
    T_true[:3, :3] = [[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]
    T_true[:3, 3] = [0.06, -0.05, 0.04]
    src = (np.linalg.inv(T_true)[:3, :3] @ scan.T).T + np.linalg.inv(T_true)[:3, 3]
    T, cov, _, valid = trk.tracking(torch.from_numpy(src.astype(np.float32)).cuda(),
                                    torch.eye(4, dtype=torch.float64, device="cuda"))
    assert valid
    T = T.cpu().numpy()
    assert abs(T[2, 3] - T_true[2, 3]) < 0.01, T  # z is observable on the (near horizontal) sheets
    assert np.abs(T[:3, :3] - T_true[:3, :3]).max() < 0.01


@pytest.mark.parametrize("wf", [True, False])
def test_grouped_iterations_train_like_single_ones(wf, monkeypatch):
    """Mapper.mapping gathers and searches a group of iterations in one launch each (their inputs do not depend on the
    training) and stages the decoder once per call; `group_iterations = False` keeps one gather / kNN per iteration.
    With the weight gradient in line, the decoder's gradient of an iteration is left as slot copies and summed by the next
    iteration's lazy-Adam launch (three launches per iteration); PIN_DEFER_DEC_REDUCE=0 keeps the reduction launch.  Grouped and in
    line, the iterations of a group are queued by one foreign call (engine.MapTrainer.step_group).
    The weight gradient and the decoder's step of an iteration run on a side stream beside the next iteration's
    optimiser launch; `overlap_weight_grad = False` keeps them in line.  Same seed, same state: the same batches, and
    the trained features / decoder agree to rounding (atomics order) whatever the launch structure."""
    from pin_slam_amd import synth
    from pin_slam_amd.dropin.model.decoder import Decoder
    from pin_slam_amd.dropin.model.neural_points import NeuralPoints
    from pin_slam_amd.dropin.utils.mapper import Mapper
    rng = np.random.default_rng(3)
    pts, _ = synth.disc_points(rng, 60_000, 20.0, 2)
    base, _ = synth.disc_points(rng, 200_000, 19.0, 2)
    nrm = synth.sheet_normal(base[:, 0].astype(np.float64), base[:, 1].astype(np.float64))
    dd = 0.15 * rng.standard_normal(len(base))
    results = []
    deferred = []
    for grouped, overlap, defer in ((True, True, "1"), (False, True, "1"), (True, False, "1"), (True, False, "0"), (False, False, "1")):
        monkeypatch.setenv("PIN_DEFER_DEC_REDUCE", defer)
        torch.manual_seed(11)
        cfg = _cfg(search_alpha=0.5, query_nn_k=8, bs=2048, local_map_radius=40.0, local_map_travel_dist_ratio=5.0,
                   weighted_first=wf)
        npts = NeuralPoints(cfg)
        npts.travel_dist = torch.zeros(1, device="cuda")
        npts.update(torch.from_numpy(pts).cuda(), torch.zeros(3), torch.eye(3), 0)
        dec = Decoder(cfg, 32, 2, 1) if wf else Decoder(cfg, 64, 1, 1)  # (per-neighbour decoding: the one-layer tile kernel)
        mp = Mapper(cfg, _FakeDataset(), npts, {"sdf": dec, "semantic": None, "color": None})
        mp.group_iterations = grouped
        mp._get_trainer().overlap_weight_grad = overlap  # weight gradient + decoder step on a side stream, or in line
        mp.global_coord_pool = torch.from_numpy((base + dd[:, None] * nrm).astype(np.float32)).cuda()
        mp.coord_pool = mp.global_coord_pool
        mp.sdf_label_pool = torch.from_numpy(dd.astype(np.float32)).cuda()
        mp.weight_pool = torch.ones(len(base), device="cuda")
        mp.time_pool = torch.zeros(len(base), dtype=torch.int, device="cuda")
        mp.pool_sample_count = len(base)
        torch.manual_seed(5)
        from pin_slam_amd import engine as _engine, ops as _ops
        seen, real = [], _ops.train_deferred_partial
        monkeypatch.setattr(_ops, "train_deferred_partial", lambda: (seen.append(real()), seen[-1])[1])
        groups, real_group = [], _engine.MapTrainer.step_group
        monkeypatch.setattr(_engine.MapTrainer, "step_group", lambda self, *a, **k: (groups.append(a[-1]), real_group(self, *a, **k))[1])
        mp.mapping(20)  # more than one group of 16
        monkeypatch.setattr(_ops, "train_deferred_partial", real)
        monkeypatch.setattr(_engine.MapTrainer, "step_group", real_group)
        deferred.append((sum(x is not None for x in seen), tuple(groups)))
        assert mp._trainer.buf.group == 16 and mp._trainer._pending_partial is None
        # assign_local_to_global copied back the rows the call changed -- which is all that differs: global == local afterwards
        assert npts.local_count() == npts.count() and npts._changed_rows is None
        n_pts = npts.count()
        touched = int((mp._trainer.lazy.state[:n_pts] != 0).sum())
        assert 0 < touched < n_pts  # (the marker is selective here: some rows were never read)
        assert torch.equal(npts.geo_features[:n_pts], npts.local_geo_features.data[:n_pts])
        assert torch.equal(npts.point_certainties[:n_pts], npts.local_point_certainties[:n_pts])
        assert torch.equal(npts.point_ts_update[:n_pts], npts.local_point_ts_update[:n_pts])
        results.append((npts.local_geo_features.data.clone(), dec.flat_params().clone(), npts.local_point_certainties.clone()))
    # the deferred reduction ran where it can: in line (no second stream), in 19 of the 20 iterations (the last one reduces itself) --
    # grouped, the whole loop behind the ABI (pin_train_group_steps: one call per group of 16 + 4), in its two-stream form (first
    # arm) and in line (third); iteration by iteration from Python otherwise
    assert deferred == [(0, (16, 4)), (0, ()), (0, (16, 4)), (0, ()), (19, ())], deferred
    fa, da, ca = results[0]
    assert not torch.equal(fa, torch.zeros_like(fa))
    for fb, db, cb in results[1:]:
        # (measured: mean |difference| 1e-8 -- only the order of the float atomics differs; Adam with eps = 1e-15 can turn
        # the rounding noise of a near-zero gradient into a step of ~lr, hence a bound on the share of such entries too)
        assert (fa - fb).abs().mean().item() < 1e-5 and ((fa - fb).abs() > 5e-3).float().mean().item() < 1e-3
        assert (da - db).abs().max().item() < 1e-3
        torch.testing.assert_close(ca, cb, rtol=1e-4, atol=1e-4)


def test_mapper_with_analytic_eikonal_term():
    """run_livox.yaml's training mode through the drop-in Mapper: per-neighbour decoding, 8 neighbours, the Eikonal term
    on the autograd gradient of every sample (numerical_grad False -> gradient_decimation 1).  Training lowers the SDF
    error and pulls the gradient norm of the field to 1; the same with weighted-first decoding."""
    from pin_slam_amd import ops, synth
    from pin_slam_amd.dropin.model.decoder import Decoder
    from pin_slam_amd.dropin.model.neural_points import NeuralPoints
    from pin_slam_amd.dropin.utils.mapper import Mapper
    torch.manual_seed(0)
    cfg = _cfg(search_alpha=0.5, query_nn_k=8, bs=4096, local_map_radius=40.0, local_map_travel_dist_ratio=5.0,
               weighted_first=False, numerical_grad=False, gradient_decimation=1, weight_e=0.5, loss_weight_on=True)
    rng = np.random.default_rng(0)
    pts, _ = synth.disc_points(rng, 120_000, 25.0, 2)
    npts = NeuralPoints(cfg)
    npts.travel_dist = torch.zeros(1, device="cuda")
    npts.update(torch.from_numpy(pts).cuda(), torch.zeros(3), torch.eye(3), 0)
    dec = Decoder(cfg, 64, 1, 1)
    mp = Mapper(cfg, _FakeDataset(), npts, {"sdf": dec, "semantic": None, "color": None})
    base, _ = synth.disc_points(rng, 400_000, 24.0, 2)
    nrm = synth.sheet_normal(base[:, 0].astype(np.float64), base[:, 1].astype(np.float64))
    dd = 0.15 * rng.standard_normal(len(base))
    mp.global_coord_pool = torch.from_numpy((base + dd[:, None] * nrm).astype(np.float32)).cuda()
    mp.coord_pool = mp.global_coord_pool
    mp.sdf_label_pool = torch.from_numpy(dd.astype(np.float32)).cuda()
    mp.weight_pool = torch.ones(len(base), device="cuda")
    mp.time_pool = torch.zeros(len(base), dtype=torch.int, device="cuda")
    mp.pool_sample_count = len(base)
    probe, lab = mp.global_coord_pool[:20000], mp.sdf_label_pool[:20000]

    def grad_norm():
        nbr, nn, _ = npts.knn(probe, True)
        fs = npts.field_state(dec, query_locally=True)
        _, g, _, _ = ops.sdf_query(fs, probe, nbr, nn, grad=True, std=False)
        return g.norm(dim=1)[nn >= 4]

    err0, n0 = (mp.sdf(probe)[0] - lab).abs().mean().item(), grad_norm()
    mp.mapping(300)
    assert mp._trainer.eikonal == "analytic" and mp._trainer.buf.n_eik == 0
    err1, n1 = (mp.sdf(probe)[0] - lab).abs().mean().item(), grad_norm()
    assert err1 < 0.35 * err0 and err1 < 0.05, (err0, err1)
    assert (n1 - 1).abs().mean().item() < 0.2 and (n1 - 1).abs().mean().item() < 0.5 * (n0 - 1).abs().mean().item()
    # weighted-first decoding with the same term (train_fused_an_kernel): the field keeps training
    cfg.weighted_first = True
    mp2 = Mapper(cfg, _FakeDataset(), npts, {"sdf": dec, "semantic": None, "color": None})
    mp2.coord_pool, mp2.global_coord_pool, mp2.sdf_label_pool = mp.coord_pool, mp.global_coord_pool, mp.sdf_label_pool
    mp2.weight_pool, mp2.time_pool, mp2.pool_sample_count = mp.weight_pool, mp.time_pool, mp.pool_sample_count
    mp2.mapping(1)
    e0 = (mp2.sdf(probe)[0] - lab).abs().mean().item()
    mp2.mapping(200)
    assert mp2._trainer.eikonal == "analytic" and mp2._trainer.fs.weighted_first
    e1 = (mp2.sdf(probe)[0] - lab).abs().mean().item()
    assert e1 < 0.05 and e1 < 1.05 * e0, (e0, e1)
    cfg.weighted_first = False


def test_pool_records_reused_across_a_mapping_call():
    """Mapper.mapping with more draws than pool samples (the C4 shape in small): the neighbour records of the drawn samples
    are copied out of ONE search over the pool (pin_gather_records_drawn) and only the Eikonal probes are searched per
    iteration -- the group buffers must hold, bit for bit, what the search over the group's queries gives."""
    from pin_slam_amd import synth
    from pin_slam_amd.dropin.model.decoder import Decoder
    from pin_slam_amd.dropin.model.neural_points import NeuralPoints
    from pin_slam_amd.dropin.utils.mapper import Mapper
    torch.manual_seed(0)
    cfg = _cfg(search_alpha=0.5, query_nn_k=8, bs=8192, local_map_radius=40.0, local_map_travel_dist_ratio=5.0)
    rng = np.random.default_rng(0)
    pts, _ = synth.disc_points(rng, 60_000, 20.0, 2)
    npts = NeuralPoints(cfg)
    npts.travel_dist = torch.zeros(1, device="cuda")
    npts.update(torch.from_numpy(pts).cuda(), torch.zeros(3), torch.eye(3), 0)
    dec = Decoder(cfg, 64, 1, 1)
    mp = Mapper(cfg, _FakeDataset(), npts, {"sdf": dec, "semantic": None, "color": None})
    base, _ = synth.disc_points(rng, 20_000, 19.0, 2)
    dd = 0.15 * rng.standard_normal(len(base))
    mp.global_coord_pool = torch.from_numpy((base + dd[:, None] * np.array([0.0, 0.0, 1.0])).astype(np.float32)).cuda()
    mp.coord_pool = mp.global_coord_pool
    mp.sdf_label_pool = torch.from_numpy(dd.astype(np.float32)).cuda()
    mp.weight_pool = torch.ones(len(base), device="cuda")
    mp.time_pool = torch.zeros(len(base), dtype=torch.int, device="cuda")
    mp.pool_sample_count = len(base)
    checked = []
    orig = mp._records_group

    def checking(t, it0, gn, rec):
        orig(t, it0, gn, rec)
        n = gn * t.buf.Q
        got_nbr, got_nn = t.buf.nbr_all[:n].clone(), t.buf.nn_all[:n].clone()
        t.knn_group(gn)  # the plain search over the same queries
        assert torch.equal(got_nn, t.buf.nn_all[:n])
        assert torch.equal(got_nbr.view(torch.int32), t.buf.nbr_all[:n].view(torch.int32))
        checked.append(gn)

    mp._records_group = checking
    mp.mapping(6)  # 6 x 8192 draws from 20 000 pool samples: above the ratio, the reuse path runs by itself
    assert sum(checked) == 6
    mp.reuse_pool_records = False
    checked.clear()
    mp.mapping(2)
    assert not checked


def test_mini_slam_loop_with_colour():
    """colour_on through the drop-in classes: Mapper.mapping trains the colour field next to the
    SDF (mapper.py:668-671, 802-812), Tracker.query_source_points returns colour + per-channel
    gradients (tracker.py:342-350) and Tracker.tracking runs with the photometric term
    (tracker.py:493-542)."""
    from pin_slam_amd import synth
    from pin_slam_amd.dropin.model.decoder import Decoder
    from pin_slam_amd.dropin.model.neural_points import NeuralPoints
    from pin_slam_amd.dropin.utils.mapper import Mapper
    from pin_slam_amd.dropin.utils.tracker import Tracker
    torch.manual_seed(0)
    cfg = _cfg(search_alpha=0.5, query_nn_k=8, bs=4096, local_map_radius=40.0, local_map_travel_dist_ratio=5.0,
               reg_iter_n=30, color_on=True, color_channel=3, photometric_loss_on=True, feature_std=0.01)
    rng = np.random.default_rng(0)
    pts, _ = synth.disc_points(rng, 120_000, 25.0, 2)
    npts = NeuralPoints(cfg)
    npts.travel_dist = torch.zeros(1, device="cuda")
    npts.update(torch.from_numpy(pts).cuda(), torch.zeros(3), torch.eye(3), 0)
    assert npts.local_color_features.shape == (npts.local_count() + 1, 8)
    decoders = {"sdf": Decoder(cfg, 64, 1, 1), "semantic": None, "color": Decoder(cfg, 64, 1, 3)}
    mp = Mapper(cfg, _FakeDataset(), npts, decoders)

    def paint(p):  # smooth RGB texture over the sheets
        x, y = p[:, 0], p[:, 1]
        return np.stack([0.5 + 0.4 * np.sin(0.7 * x), 0.5 + 0.4 * np.cos(0.5 * y), 0.5 + 0.3 * np.sin(0.4 * (x + y))], 1).astype(np.float32)

    base, _ = synth.disc_points(rng, 400_000, 24.0, 2)
    nrm = synth.sheet_normal(base[:, 0].astype(np.float64), base[:, 1].astype(np.float64))
    dd = 0.15 * rng.standard_normal(len(base))
    mp.global_coord_pool = torch.from_numpy((base + dd[:, None] * nrm).astype(np.float32)).cuda()
    mp.coord_pool = mp.global_coord_pool
    mp.sdf_label_pool = torch.from_numpy(dd.astype(np.float32)).cuda()
    mp.color_pool = torch.from_numpy(paint(base)).cuda()
    mp.weight_pool = torch.ones(len(base), device="cuda")
    mp.time_pool = torch.zeros(len(base), dtype=torch.int, device="cuda")
    mp.pool_sample_count = len(base)
    trk = Tracker(cfg, npts, decoders)
    probe = torch.from_numpy(base[:20000]).cuda()
    target = torch.from_numpy(paint(base[:20000])).cuda()
    col0 = trk.query_source_points(probe, cfg.infer_bs, query_color=True)[2]
    mp.mapping(300)
    sdf, grad, col1, cgrad, _, mask, cert, std = trk.query_source_points(probe, cfg.infer_bs, query_color=True,
                                                                         query_color_grad=True)
    e0, e1 = (col0 - target).abs().mean().item(), (col1 - target).abs().mean().item()
    assert e1 < 0.4 * e0 and e1 < 0.05, (e0, e1)
    assert cgrad.shape == (20000, 3, 3) and torch.isfinite(cgrad[mask]).all()
    assert torch.allclose(npts.color_features[:-1], npts.local_color_features.data[:-1])
    # regress_color on the queried colour features == the fused colour query (tensor API vs kernel)
    _, cfeat, w, nn, _ = npts.query_feature(probe, training_mode=False, query_geo_feature=False, query_color_feature=True)
    np.testing.assert_allclose(decoders["color"].regress_color(cfeat).cpu().numpy(), col1.cpu().numpy(), rtol=1e-4, atol=1e-5)
    scan, _ = synth.disc_points(rng, 20_000, 20.0, 2)
    T_true = np.eye(4)
    T_true[:3, 3] = [0.06, -0.05, 0.04]
    src = scan - T_true[:3, 3]
    T, cov, _, valid = trk.tracking(torch.from_numpy(src.astype(np.float32)).cuda(),
                                    torch.eye(4, dtype=torch.float64, device="cuda"),
                                    source_colors=torch.from_numpy(paint(scan)).cuda())
    assert valid
    T = T.cpu().numpy()
    assert abs(T[2, 3] - T_true[2, 3]) < 0.01, T
    # the colour texture makes the in-plane translation observable as well
    assert np.abs(T[:2, 3] - T_true[:2, 3]).max() < 0.03, T


@pytest.mark.parametrize("case", ["c2_wf", "kitti_nwf"])
@pytest.mark.parametrize("local", [False, True, "global_bricks"])
def test_mesher_query_points_matches_reference(case, local):
    """Drop-in Mesher.query_points (fused search + decode, chunked) against the reference's output on a grid
    that reaches into unobserved space: mask identical, SDF within 1e-4, zeros where nothing is near.
    "global_bricks": the call builds a brick cache over the global map for its searches (what it does from 3e6 queries on)."""
    use_bricks, local = local == "global_bricks", local is True
    import ctypes as C
    from pin_slam_amd import _lib, ops
    from pin_slam_amd.dropin.utils.mesher import Mesher
    from tests import gpu_util as U
    d, mz = G.load(case), G.load("mesher")
    st = U.search_state(d)
    d2 = dict(d)
    d2["geo_features"], d2["local_geo_features"], d2["dec_flat"] = (mz[case + "_geo_features"], mz[case + "_local_geo_features"],
                                                                   mz[case + "_dec_flat"])

    class _MapCfg:
        num_nei_cells = int(d["num_nei_cells"])

    class _Pts:  # what Mesher.query_points touches on the map object
        config = _MapCfg()
        neighbor_dx = torch.from_numpy(np.ascontiguousarray(d["neighbor_dx"]))
        neighbor_K = int(d["neighbor_dx"].shape[0])

        def knn(self, q, query_locally):
            assert not use_bricks, "the call should search through its own brick cache"
            return ops.knn_query(st, q, int(d["query_nn_k"]), time_filtering=query_locally, local=query_locally)

        def field_state(self, decoder, query_locally=True, color=False):
            return U.field_state(d2, local=query_locally)

        def count(self):
            return int(st.n_points)

        def search_state(self):
            return st

        def _wait_bricks(self):
            pass

    class _Cfg:
        silence, device, dtype, color_channel, query_nn_k = True, "cuda", torch.float32, 0, int(d["query_nn_k"])

    mesher = Mesher(_Cfg(), _Pts(), {"sdf": None, "semantic": None, "color": None})
    if use_bricks:
        mesher.global_bricks_min_queries = 0
    grid = torch.from_numpy(mz[case + "_grid"]).cuda()
    sdf, _, _, mask = mesher.query_points(grid, 3000, True, False, False, True, query_locally=local, out_torch=True)
    key = "local" if local else "global"
    assert np.array_equal(mask.numpy() != 0, mz[f"{case}_mask_{key}"] != 0)
    np.testing.assert_allclose(sdf.numpy(), mz[f"{case}_sdf_{key}"], rtol=1e-4, atol=3e-6)
    assert (getattr(mesher, "_global_bricks", None) is not None) == use_bricks
    mesher.global_bricks_min_queries = 10 ** 9
    _Pts.knn = lambda self, q, query_locally: ops.knn_query(st, q, int(d["query_nn_k"]), time_filtering=query_locally, local=query_locally)
    sdf_np, _, _, mask_np = mesher.query_points(grid[:500], 200, out_torch=False)
    assert sdf_np.dtype == np.float64 and sdf_np.shape == (500,) and mask_np.shape == (500,)


def _postloop_map(pl, cfg):
    """A drop-in NeuralPoints holding the map of the postloop fixture."""
    from pin_slam_amd.dropin.model.neural_points import NeuralPoints
    npts = NeuralPoints(cfg)
    npts.travel_dist = torch.from_numpy(pl["travel_dist"]).cuda()
    P = len(pl["neural_points"])
    npts._alloc(P + 64)  # the arrays are written directly: replaying update() would depend on how the reference's
    g = npts._g          # multi-threaded index_put_ happened to resolve colliding hash slots
    g["pos"][:P] = torch.from_numpy(pl["neural_points"]).cuda()
    g["orient"][:P] = torch.from_numpy(pl["point_orientations"]).cuda()
    g["ts_create"][:P] = torch.from_numpy(pl["point_ts_create"]).cuda()
    g["ts_update"][:P] = torch.from_numpy(pl["point_ts_update"]).cuda()
    g["cert"][:P] = torch.from_numpy(pl["point_certainties"]).cuda()
    g["geo"][:P + 1] = torch.from_numpy(pl["geo_features"]).cuda()
    npts._n = P
    npts._rebuild_mirror()
    return npts


def test_post_loop_maintenance_matches_reference():
    """adjust_map, recreate_hash (both selection modes), prune_map (local / global) and transform_data_pool of the
    drop-in classes against the reference fixture."""
    from pin_slam_amd.dropin.model.decoder import Decoder
    from pin_slam_amd.dropin.utils.mapper import Mapper
    pl = G.load("postloop")
    cfg = _cfg(buffer_size=40009, local_map_radius=20.0, local_map_travel_dist_ratio=1.0, feature_std=0.05)
    B = 40009
    # prune_map on two copies
    for mode in ("local", "global"):
        npts = _postloop_map(pl, cfg)
        npts.cur_ts = 3
        assert npts.prune_map(1.0, min_prune_count=50, global_prune=mode == "global") == bool(pl[f"prune_{mode}_changed"])
        assert np.array_equal(npts.neural_points.cpu().numpy(), pl[f"prune_{mode}_points"])
        assert np.array_equal(npts.point_ts_create.cpu().numpy(), pl[f"prune_{mode}_ts_create"])
        assert np.array_equal(npts.geo_features.cpu().numpy(), pl[f"prune_{mode}_geo"])
    npts = _postloop_map(pl, cfg)
    pd = torch.from_numpy(pl["pose_diff"]).cuda()
    npts.adjust_map(pd)
    assert npts.after_pgo
    np.testing.assert_allclose(npts.neural_points.cpu().numpy(), pl["adj_points"], rtol=0, atol=4e-6)
    np.testing.assert_allclose(npts.point_orientations.cpu().numpy(), pl["adj_orient"], rtol=0, atol=1e-6)
    # continue from the reference's adjusted positions (voxel membership is discontinuous in the last bit)
    npts._g["pos"][:npts.count()] = torch.from_numpy(pl["adj_points"]).cuda()
    for mode in ("ts", "cert"):
        npts.recreate_hash(None, None, True, mode == "ts", 3)
        tab = npts.buffer_pt_index.cpu().numpy().astype(np.int64)
        ref_tab, sel = O.recreate_hash(pl["adj_points"], pl["point_ts_create"], 3, pl["resolution"], B,
                                       certainties=pl["point_certainties"], with_ts=mode == "ts")
        slots = np.nonzero(tab >= 0)[0]
        assert np.array_equal(slots, pl[f"rehash_{mode}_slots"])
        writers = np.bincount(O.hash_slots(O.grid_coords(pl["adj_points"][sel], pl["resolution"]), B), minlength=B)
        ref = np.full(B, -1, np.int64); ref[pl[f"rehash_{mode}_slots"]] = pl[f"rehash_{mode}_vals"]
        assert np.array_equal(tab[writers == 1], ref[writers == 1])   # single-writer slots: the reference's table
        cand = np.isin(tab[writers > 1], sel)                        # colliding slots: one of the voxel winners
        assert cand.all()
    # Mapper.transform_data_pool
    mp = Mapper(cfg, _FakeDataset(), npts, {"sdf": Decoder(cfg, 32, 1, 1), "semantic": None, "color": None})
    n = len(pl["pool_ts"])
    mp.coord_pool = torch.from_numpy(pl["pool_global"]).cuda()
    mp.global_coord_pool = torch.from_numpy(pl["pool_global"]).cuda()
    mp.sdf_label_pool = torch.zeros(n, device="cuda"); mp.weight_pool = torch.ones(n, device="cuda")
    mp.time_pool = torch.from_numpy(pl["pool_ts"]).cuda()
    mp.pool_sample_count = n
    mp.transform_data_pool(pd)
    np.testing.assert_allclose(mp.global_coord_pool.cpu().numpy(), pl["pool_global_after"], rtol=0, atol=4e-6)
    assert np.array_equal(mp.coord_pool.cpu().numpy(), pl["pool_global"])


def test_final_merge_and_mid_ts_match_reference():
    """recreate_hash(None, None, False, False) after prune_map(thre, 0, True) -- the end of every pin_slam.py run
    (pin_slam.py:520-521) -- and the use_mid_ts variants of adjust_map / recreate_hash, on the device kernels
    (pin_prune_map, pin_hash_rebuild), against the reference fixture."""
    import dataclasses
    pl = G.load("postloop")
    B = 40009
    cfg = _cfg(buffer_size=B, local_map_radius=20.0, local_map_travel_dist_ratio=1.0, feature_std=0.05)
    npts = _postloop_map(pl, cfg)
    npts.cur_ts = 3
    npts._g["pos"][:npts.count()] = torch.from_numpy(pl["adj_points"]).cuda()
    npts._g["orient"][:npts.count()] = torch.from_numpy(pl["adj_orient"]).cuda()
    assert npts.prune_map(1.0, 0, True)
    npts.recreate_hash(None, None, False, False)
    n = npts.count()
    assert n == len(pl["merge_points"]) and n < 0.9 * len(pl["neural_points"])
    for got, key in ((npts.neural_points, "merge_points"), (npts.point_orientations, "merge_orient"),
                     (npts.geo_features, "merge_geo"), (npts.point_ts_create, "merge_ts_create"),
                     (npts.point_ts_update, "merge_ts_update"), (npts.point_certainties, "merge_cert")):
        assert np.array_equal(got.cpu().numpy(), pl[key]), key
    tab = npts.buffer_pt_index.cpu().numpy().astype(np.int64)
    assert np.array_equal(np.nonzero(tab >= 0)[0], pl["merge_slots"])
    ref = np.full(B, -1, np.int64); ref[pl["merge_slots"]] = pl["merge_vals"]
    slot_of = O.hash_slots(O.grid_coords(pl["merge_points"], pl["resolution"]), B)
    writers = np.bincount(slot_of, minlength=B)
    assert np.array_equal(tab[writers == 1], ref[writers == 1])
    for sl in np.nonzero(writers > 1)[0][:200]:  # colliding slots keep the last writer in index order
        assert tab[sl] == np.nonzero(slot_of == sl)[0][-1]
    # the search mirror was rebuilt: a kNN query over the merged map answers with merged indices
    npts.reset_local_map(torch.tensor([13.0, 0.0, 0.0]), None, 3)
    pos4 = npts._g["pos4"][:n].cpu().numpy()
    assert np.array_equal(pos4[:, :3], pl["merge_points"]) and np.array_equal(pos4[:, 3].view(np.int32), pl["merge_ts_create"])
    # use_mid_ts
    cfg2 = dataclasses.replace(cfg, use_mid_ts=True) if dataclasses.is_dataclass(cfg) else cfg
    if cfg2 is cfg:
        cfg.use_mid_ts = True
    q = _postloop_map(pl, cfg2)
    q.adjust_map(torch.from_numpy(pl["pose_diff"]).cuda())
    np.testing.assert_allclose(q.neural_points.cpu().numpy(), pl["mid_adj_points"], rtol=0, atol=4e-6)
    np.testing.assert_allclose(q.point_orientations.cpu().numpy(), pl["mid_adj_orient"], rtol=0, atol=1e-6)
    q._g["pos"][:q.count()] = torch.from_numpy(pl["mid_adj_points"]).cuda()
    q.recreate_hash(None, None, True, True, 3)
    tab = q.buffer_pt_index.cpu().numpy().astype(np.int64)
    assert np.array_equal(np.nonzero(tab >= 0)[0], pl["rehash_mid_slots"])
    _, sel = O.recreate_hash(pl["mid_adj_points"], pl["point_ts_create"], 3, pl["resolution"], B, ts_update=pl["point_ts_update"])
    writers = np.bincount(O.hash_slots(O.grid_coords(pl["mid_adj_points"][sel], pl["resolution"]), B), minlength=B)
    ref = np.full(B, -1, np.int64); ref[pl["rehash_mid_slots"]] = pl["rehash_mid_vals"]
    assert np.array_equal(tab[writers == 1], ref[writers == 1])
    cfg.use_mid_ts = False


def test_local_map_variants_match_reference():
    """reset_local_map by a window of frames (pin_slam.py:287, loop_local_map_by_travel_dist = False), with a float64
    sensor position (dataset.cur_pose_torch) and with use_mid_ts, against the `update` fixture."""
    from pin_slam_amd.dropin.model.neural_points import NeuralPoints
    d = G.load("update")
    cfg = _cfg(buffer_size=int(d["buffer_size"]), local_map_radius=float(d["local_map_radius"]), local_map_travel_dist_ratio=1.0)
    npts = NeuralPoints(cfg)
    npts.travel_dist = torch.from_numpy(d["travel_dist"]).cuda()
    P = len(d["neural_points"])
    npts._alloc(P + 64)
    g = npts._g
    g["pos"][:P] = torch.from_numpy(d["neural_points"]).cuda()
    g["ts_create"][:P] = torch.from_numpy(d["point_ts_create"]).cuda()
    g["ts_update"][:P] = torch.from_numpy(d["point_ts_update"]).cuda()
    g["cert"][:P] = 0
    g["orient"][:P] = 0
    npts._n = P
    npts._rebuild_mirror()
    sp64 = torch.from_numpy(d["var_sensor"])
    npts.reset_local_map(sp64.float(), None, 2, False, 2)
    assert np.array_equal(npts.local_mask.cpu().numpy(), d["var_ts_mask"])
    assert np.array_equal(npts.global2local.cpu().numpy(), d["var_ts_g2l"])
    npts.reset_local_map(sp64, None, 3, True)
    assert np.array_equal(npts.local_mask.cpu().numpy(), d["var_f64_mask"])
    npts.reset_local_map(sp64.float(), None, 3, True)
    assert np.array_equal(npts.local_mask.cpu().numpy(), d["var_f32_mask"])
    cfg.use_mid_ts = True
    npts.reset_local_map(sp64.float(), None, 3, True)
    assert np.array_equal(npts.local_mask.cpu().numpy(), d["var_mid_mask"])
    npts.reset_local_map(sp64.float(), None, 2, False, 1)
    assert np.array_equal(npts.local_mask.cpu().numpy(), d["var_mid_ts_mask"])
    cfg.use_mid_ts = False


def test_map_pickles_like_the_reference_saves_it():
    """tools.py:295-317 saves {"neural_points": <module>, "sdf": state_dict}: the drop-in module must pickle
    (no ctypes handles in its state) and come back usable after recreate_hash, as vis_pin_map.py does."""
    import io
    from pin_slam_amd.dropin.model.decoder import Decoder
    from pin_slam_amd.dropin.model.neural_points import NeuralPoints
    d = G.load("update")
    cfg = _cfg(search_alpha=0.5, query_nn_k=8, feature_std=0.1)
    npts = NeuralPoints(cfg)
    npts.travel_dist = torch.from_numpy(d["travel_dist"]).cuda()
    for ts in range(2):
        npts.update(torch.from_numpy(d[f"pts{ts}"]).cuda(), torch.tensor([9.0 * ts, 0.0, 0.0]), torch.eye(3), ts)
    dec = Decoder(cfg, 32, 2, 1)
    q = torch.from_numpy(d["pts1"][:300] + 0.02).cuda()
    feat0, _, _, nn0, _ = npts.query_feature(q, training_mode=False, query_locally=False)
    sdf0 = dec.sdf(feat0)
    n0 = npts.count()
    npts.clear_temp()
    buf = io.BytesIO()
    torch.save({"neural_points": npts, "sdf": dec.state_dict()}, buf)
    buf.seek(0)
    loaded = torch.load(buf, weights_only=False)
    npts2 = loaded["neural_points"]
    dec2 = Decoder(cfg, 32, 2, 1)
    dec2.load_state_dict(loaded["sdf"])
    assert npts2.count() == n0
    npts2.recreate_hash(None, None, True, False, 1)  # vis_pin_map.py: rebuild the table by certainty
    feat1, _, _, nn1, _ = npts2.query_feature(q, training_mode=False, query_locally=False)
    assert torch.equal(nn0, nn1)
    np.testing.assert_allclose(dec2.sdf(feat1).cpu().numpy(), sdf0.cpu().numpy(), rtol=1e-6, atol=1e-7)


def _small_mapper(bs=2048, **kw):
    """A drop-in Mapper over a small synthetic map with a pool (the set-up of test_grouped_iterations_train_like_single_ones)."""
    from pin_slam_amd import synth
    from pin_slam_amd.dropin.model.decoder import Decoder
    from pin_slam_amd.dropin.model.neural_points import NeuralPoints
    from pin_slam_amd.dropin.utils.mapper import Mapper
    rng = np.random.default_rng(3)
    pts, _ = synth.disc_points(rng, 30_000, 20.0, 2)
    base, _ = synth.disc_points(rng, 60_000, 19.0, 2)
    nrm = synth.sheet_normal(base[:, 0].astype(np.float64), base[:, 1].astype(np.float64))
    dd = 0.15 * rng.standard_normal(len(base))
    torch.manual_seed(11)
    cfg = _cfg(search_alpha=0.5, query_nn_k=8, bs=bs, local_map_radius=40.0, local_map_travel_dist_ratio=5.0, **kw)
    npts = NeuralPoints(cfg)
    npts.travel_dist = torch.zeros(1, device="cuda")
    npts.update(torch.from_numpy(pts).cuda(), torch.zeros(3), torch.eye(3), 0)
    dec = Decoder(cfg, 32, 2, 1)
    mp = Mapper(cfg, _FakeDataset(), npts, {"sdf": dec, "semantic": None, "color": None})
    mp.global_coord_pool = torch.from_numpy((base + dd[:, None] * nrm).astype(np.float32)).cuda()
    mp.coord_pool = mp.global_coord_pool
    mp.sdf_label_pool = torch.from_numpy(dd.astype(np.float32)).cuda()
    mp.weight_pool = torch.ones(len(base), device="cuda")
    mp.time_pool = torch.zeros(len(base), dtype=torch.int, device="cuda")
    mp.pool_sample_count = len(base)
    return mp, npts, dec


def test_frozen_decoder_mapping_behind_the_abi(monkeypatch):
    """freeze_decoders (utils/tools.py:263-292: every frame of a run after freeze_after_frame) -- Mapper.mapping then trains the
    features only: no rider in the lazy launch, no weight gradient.  The group call (pin_train_group_steps with dense.param NULL)
    trains like the Python loop, and the decoder does not move."""
    from pin_slam_amd import engine as _engine
    results = []
    for group in ("1", "0"):
        monkeypatch.setenv("PIN_TRAIN_GROUP", group)
        mp, npts, dec = _small_mapper()
        for p_ in dec.parameters():
            p_.requires_grad_(False)
        dec0, feats0 = dec.flat_params().clone(), npts.local_geo_features.data.clone()
        calls, real = [], _engine.MapTrainer.step_group
        monkeypatch.setattr(_engine.MapTrainer, "step_group", lambda self, *a, **k: (calls.append(a[-1]), real(self, *a, **k))[1])
        torch.manual_seed(5)
        mp.mapping(6)
        monkeypatch.setattr(_engine.MapTrainer, "step_group", real)
        assert calls == ([6] if group == "1" else []), calls
        assert not mp._trainer.train_decoder
        assert torch.equal(dec.flat_params(), dec0) and not torch.equal(npts.local_geo_features.data, feats0)
        results.append(npts.local_geo_features.data.clone())
    fa, fb = results
    assert (fa - fb).abs().mean().item() < 1e-5 and ((fa - fb).abs() > 5e-3).float().mean().item() < 1e-3


def test_spatial_mapper_refuses_per_iteration_draws(monkeypatch):
    """PIN_DRAW_PER_ITERATION=1 (a replayed random stream: scripts/e2e_pin_slam.py --replay-draws) makes Mapper._draw_all hand
    back nothing; the one-GPU path then draws per iteration, the spatially sharded path plans its shards from the draws of
    the whole call and must say so instead of silently training on nothing."""
    from pin_slam_amd import collective
    mp, npts, dec = _small_mapper()
    mp.dp_comm, mp.dp_rank, mp.dp_world = collective.NullComm(0, 1), 0, 1  # one rank of the spatial mapper, identity exchange
    before = npts.local_geo_features.data.clone()
    monkeypatch.setenv("PIN_DRAW_PER_ITERATION", "1")
    with pytest.raises(NotImplementedError, match="PIN_DRAW_PER_ITERATION"):
        mp.mapping(3)
    monkeypatch.delenv("PIN_DRAW_PER_ITERATION")
    torch.manual_seed(5)
    mp.mapping(3)  # ... and the same object trains normally afterwards
    assert not torch.equal(before, npts.local_geo_features.data)


def test_an_aborted_call_leaves_no_owed_steps_behind():
    """A Mapper.mapping that ends between step_batch and finish_optimizer (an exception in the caller's loop) leaves the side
    stream's hand-over state set; the next call's reset_optimizer must order itself behind that stream and drop the owed
    steps -- they belong to the optimiser state that is thrown away -- and then train like a call on a fresh trainer."""
    results = []
    for abort in (False, True):
        mp, npts, dec = _small_mapper()
        t = mp._get_trainer()
        t.overlap_weight_grad = True  # the weight gradient + decoder step of an iteration on the side stream
        if abort:
            calls = {"n": 0}
            step = t.step_batch

            def failing(*a, **k):
                step(*a, **k)
                calls["n"] += 1
                if calls["n"] == 2:
                    raise RuntimeError("interrupted")
            t.step_batch = failing
            group = t.step_group

            def failing_group(*a, **k):  # (the same interruption when the iterations are queued by one foreign call)
                group(*a, **k)
                raise RuntimeError("interrupted")
            t.step_group = failing_group
            feats0, dec0 = npts.local_geo_features.data.clone(), dec.flat_params().clone()
            cert0 = npts.local_point_certainties.clone()
            torch.manual_seed(7)
            with pytest.raises(RuntimeError, match="interrupted"):
                mp.mapping(6)
            assert t._wg_pending  # the hand-over state of the interrupted iteration is still there
            t.step_batch, t.step_group = step, group
            # back to the state the other run starts from (the aborted call has moved features, decoder and certainties)
            torch.cuda.synchronize()
            npts.local_geo_features.data.copy_(feats0)
            dec.flat_params().copy_(dec0)
            npts.local_point_certainties.copy_(cert0)
        torch.manual_seed(5)
        mp.mapping(4)
        assert not t._wg_pending and t._dp_pending is None
        results.append((npts.local_geo_features.data.clone(), dec.flat_params().clone()))
    (fa, da), (fb, db) = results
    assert (fa - fb).abs().mean().item() < 1e-5 and (da - db).abs().max().item() < 1e-3


def test_an_aborted_spatial_call_still_takes_the_exchanged_steps():
    """a spatially sharded Mapper.mapping aborted on ONE rank (an exception between step_batch and finish_optimizer)
    leaves the last iteration's halo / decoder steps pending behind the side stream's all-reduce.  The other ranks take those
    steps; reset_optimizer of the next call must take them here as well (engine.MapTrainer._dp_finish_exchange) instead of
    discarding them -- otherwise the replicas of halo rows and decoder diverge silently.  One rank with the identity exchange:
    the decoder after the aborted call's reset is the decoder of an uninterrupted call stopped at the same iteration."""
    from pin_slam_amd import collective
    decs = []
    for abort in (False, True):
        mp, npts, dec = _small_mapper()
        mp.dp_comm, mp.dp_rank, mp.dp_world = collective.NullComm(0, 1), 0, 1
        t = mp._get_trainer()
        assert t.dp is not None and t.overlap_exchange
        calls = {"n": 0}
        step = t.step_batch

        def counting(*a, _step=step, **k):
            _step(*a, **k)
            calls["n"] += 1
            if abort and calls["n"] == 2:
                raise RuntimeError("interrupted")
        t.step_batch = counting
        torch.manual_seed(7)
        if abort:
            with pytest.raises(RuntimeError, match="interrupted"):
                mp.mapping(6)
            assert t._dp_pending is not None  # iteration 2's halo / decoder steps are still owed
            t.step_batch = step
            t.reset_optimizer(3)  # what the next mapping() call starts with
            assert t._dp_pending is None
        else:
            mp.mapping(2)  # the same two iterations, finished properly (finish_optimizer takes the owed steps)
        torch.cuda.synchronize()
        decs.append(dec.flat_params().clone())
    # (float atomics order the gradient sums differently from run to run: 1e-8; a dropped step would be a whole Adam step, ~lr = 1e-2)
    assert (decs[0] - decs[1]).abs().max().item() < 1e-5, (decs[0] - decs[1]).abs().max().item()


def test_decoder_outside_the_fp16_range_raises(monkeypatch):
    """a decoder parameter the split-fp16 image cannot hold (|w| >= 65504 or non-finite) must end in an
    exception -- raised from the status word the staging kernel sets and the registration loop carries in its one read-back --
    not in NaNs in a pose.  PIN_MLP=f32 is the advertised way out; the flag is cleared by the raise."""
    from pin_slam_amd import ops
    from pin_slam_amd.dropin.utils.tracker import Tracker
    if os.environ.get("PIN_MLP", "") == "f32":
        pytest.skip("fp32 image: no fp16 range to leave")
    mp, npts, dec = _small_mapper()
    trk = Tracker(mp.config, npts, {"sdf": dec, "semantic": None, "color": None})
    src = npts.neural_points[:5000].clone()
    T0 = torch.eye(4, dtype=torch.float64, device="cuda")
    ops.status(clear=True)
    trk.tracking(src, T0)  # fine
    assert ops.status() == 0
    keep = dec.layers[0].weight.data[3, 2].item()
    dec.layers[0].weight.data[3, 2] = 7.0e4
    with pytest.raises(RuntimeError, match="PIN_MLP=f32"):
        trk.tracking(src, T0)
    assert ops.status() == 0  # cleared by the raise
    dec.layers[0].weight.data[3, 2] = float("nan")
    with pytest.raises(RuntimeError, match="PIN_MLP=f32"):
        trk.tracking(src, T0)
    dec.layers[0].weight.data[3, 2] = keep
    T, _, _, ok = trk.tracking(src, T0)  # ... and the same objects work again
    assert torch.isfinite(T).all()


def test_registration_residual_divided_by_the_gradient_norm():
    """reg_dist_div_grad_norm (tracker.py:452-456; off in every shipped configuration): the residual of a valid point is
    sdf / |grad|, the Jacobian rows and the gradient-anomaly weight keep the plain values.  Tile kernel (weighted-first) and
    the per-neighbour tile kernel against the oracle's step on the kernels' own per-point outputs."""
    from oracle import pin_oracle as O
    from pin_slam_amd import ops
    from pin_slam_amd._lib import GnParams
    from tests import golden_util as G, gpu_util as U
    for case in ("c2_wf", "kitti_nwf"):
        d = G.load(case)
        st, fs = U.search_state(d), U.field_state(d)
        src = U.dev(d["reg_cur"])
        k = int(d["query_nn_k"])
        nbr, nn, _ = ops.knn_query(st, src, k)
        gp = GnParams()
        gp.valid_nn_k = int(d["track_mask_query_nn_k"])
        gp.min_grad_norm, gp.max_grad_norm = d["cfg_reg_min_grad_norm"], d["cfg_reg_max_grad_norm"]
        gp.max_sdf_std = d["cfg_surface_sample_range_m"] * d["cfg_max_sdf_std_ratio"]
        gp.gm_dist, gp.gm_grad = d["cfg_reg_GM_dist_m"], d["cfg_reg_GM_grad"]
        sdf, grad, std, _ = ops.sdf_query(fs, src, nbr, nn)
        outs = []
        for flag in (0, 1):
            gp.dist_div_grad_norm = flag
            sums, _, _ = ops.gn_accumulate(fs, gp, src, nbr, nn)
            T, cnt, res_cm, _ = ops.solve_gn(sums.cpu().numpy(), d["cfg_reg_lm_lambda"])
            ref = O.registration_step(d["reg_cur"], sdf.cpu().numpy(), grad.cpu().numpy(), std.cpu().numpy(), nn.cpu().numpy(),
                                      valid_nn_k=gp.valid_nn_k, min_grad_norm=gp.min_grad_norm, max_grad_norm=gp.max_grad_norm,
                                      max_sdf_std=gp.max_sdf_std, GM_dist=gp.gm_dist, GM_grad=gp.gm_grad,
                                      lm_lambda=d["cfg_reg_lm_lambda"], dist_div_grad_norm=bool(flag))
            assert abs(cnt - ref["valid_count"]) <= 2
            np.testing.assert_allclose(T, ref["T"], rtol=0, atol=1e-5)
            assert abs(res_cm - ref["residual_cm"]) < 2e-3 * max(1.0, ref["residual_cm"])
            outs.append(T)
        assert np.abs(outs[0] - outs[1]).max() > 1e-6, "the switch changes the step"
