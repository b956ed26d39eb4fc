"""GPU parity of the semantic head (config.semantic_on, run_demo_sem.yaml) through the C ABI: csrc/sem.h against the `variants` /
`process_sem` fixtures recorded from the unmodified reference (oracle/make_golden.py gen_variants / gen_process_sem, on the maps
of c2_wf -- weighted-first, 2 x 32 -- and kitti_nwf -- per-neighbour, 1 x 64) and against the oracle on further inputs."""
import dataclasses

import numpy as np
import pytest
import torch

from oracle import pin_oracle as O
from tests import golden_util as G

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["c2_wf", "kitti_nwf"])
def sg(request):
    from tests import gpu_util as U
    d = G.load(request.param)
    v = G.load("variants")
    d["name"] = request.param
    d["v"] = {k[len(request.param) + 1:]: x for k, x in v.items() if k.startswith(request.param + "_")}
    d["S"] = int(d["v"]["sem_heads"])
    d["table"] = G.dense_table(d)
    d["st"], d["fs"] = U.search_state(d), U.field_state(d)
    d["fsem"] = dataclasses.replace(d["fs"], dec=U.dev(d["v"]["sem_dec_flat"]), out_dim=d["S"])
    d["sparams"] = O.unpack_decoder(d["v"]["sem_dec_flat"], 11, int(d["dec_hidden"]), int(d["dec_levels"]), out_dim=d["S"])
    return d


def _search(d, q):
    return O.radius_search(q, d["table"], d["neural_points"], d["resolution"], d["neighbor_dx"], d["max_valid_dist2"],
                           ts_create=d["point_ts_create"], travel_dist=d["travel_dist"], cur_ts=int(d["cur_ts"]),
                           diff_travel_dist_local=d["diff_travel_dist_local"])


def test_decoder_sem_label_prob(sg):
    """Decoder.sem_label_prob on the reference's own query features (pin_decoder_sem): log-probabilities within 1e-4, rows
    that sum to one, and the raw mlp() outputs."""
    from pin_slam_amd import ops
    from tests import gpu_util as U
    d, v = sg, sg["v"]
    feat = d["qf_loc_feat"].reshape(-1, 11)
    lp = ops.decoder_sem(d["fsem"], U.dev(feat), d["S"]).cpu().numpy()
    ref = v["sem_prob"].reshape(-1, d["S"])
    np.testing.assert_allclose(lp, ref, rtol=1e-4, atol=2e-5)
    assert np.allclose(np.exp(lp.astype(np.float64)).sum(-1), 1.0, atol=1e-5)
    raw = ops.decoder_sem(d["fsem"], U.dev(feat), d["S"], raw=True).cpu().numpy()
    np.testing.assert_allclose(raw, O.mlp_forward(feat.astype(np.float64), d["sparams"]), rtol=1e-4, atol=2e-5)


def test_sem_query_labels(sg):
    """Tracker.query_source_points(query_sem=True): the argmax of the (weighted) log-probabilities equals the reference's label
    wherever its two best classes are not within the arithmetic noise; the log-probabilities against the oracle."""
    from pin_slam_amd import ops
    from tests import gpu_util as U
    d, v = sg, sg["v"]
    k, wf = int(d["query_nn_k"]), bool(d["weighted_first"])
    q = U.dev(d["query"])
    nbr, nn, _ = ops.knn_query(d["st"], q, k)
    lab, lp = ops.sem_query(d["fsem"], q, nbr, nn, d["S"], want_logprob=True)
    pred, olab, _ = O.query_sem(d["query"], _search(d, d["query"]), d["local_geo_features"], d["local_neural_points"], d["sparams"], k,
                                weighted_first=wf, global2local=d["global2local"])
    np.testing.assert_allclose(lp.cpu().numpy(), pred, rtol=1e-4, atol=3e-5)
    top2 = np.sort(pred, -1)[:, -2:]
    clear = (top2[:, 1] - top2[:, 0]) > 1e-4
    assert clear.mean() > 0.98
    assert np.array_equal(lab.cpu().numpy()[clear], v["sem_pred"].astype(np.int32)[clear])
    assert np.array_equal(lab.cpu().numpy()[clear], olab[clear])


def test_sem_select_matches_the_reference_rule():
    from pin_slam_amd import ops
    rng = np.random.default_rng(3)
    for n in (1, 63, 1024, 5000, 70001):
        lab = rng.integers(-1, 21, n).astype(np.int32)
        for fs_on in (False, True):
            for dec in (1, 3, 10):
                sel, cnt = ops.sem_select(torch.from_numpy(lab).cuda(), fs_on, dec)
                ref = O.sem_select_mask(lab, fs_on, dec)
                assert np.array_equal(sel.cpu().numpy().astype(bool), ref), (n, fs_on, dec)
                assert int(cnt.item()) == int(ref.sum())


def test_mapping_two_iterations_with_the_semantic_term(sg):
    """pin_train_step + pin_train_sem_step on the fixture's fixed batches: the geometry-feature gradient (SDF + semantic term),
    the SDF decoder's and the semantic decoder's gradients against the reference's autograd, the NLL term and the total loss,
    the parameters after the two Adam steps."""
    from pin_slam_amd import ops, sharding
    from tests import gpu_util as U
    d, v = sg, sg["v"]
    k, H, L, S = int(d["query_nn_k"]), int(d["dec_hidden"]), int(d["dec_levels"]), sg["S"]
    wf = bool(d["weighted_first"])
    feats, dec, sdec = U.dev(d["local_geo_features"]), U.dev(d["dec_flat"]), U.dev(v["sem_dec_flat"])
    cert, tsu = U.dev(d["local_point_certainties"]), U.dev(d["local_point_ts_update"], torch.int32)
    fs = dataclasses.replace(d["fs"], feats=feats, dec=dec, certainty=cert)
    fsem = dataclasses.replace(d["fs"], feats=feats, dec=sdec, certainty=None, out_dim=S)
    gfeat, gdec, gsem = torch.zeros_like(feats), torch.zeros_like(dec), torch.zeros_like(sdec)
    state = [torch.zeros_like(t) for t in (feats, feats, dec, dec, sdec, sdec)]
    bs = v["map_coord0"].shape[0]
    dec_n = int(v["cfg_gradient_decimation"])
    buf = ops.TrainBuffers(bs, dec_n, k, H, L, weighted_first=wf)
    n_eik = sharding.n_eik_global(bs, dec_n)
    lr, aeps = float(v["cfg_lr"]), float(v["cfg_adam_eps"])
    gfs, gds, gss = [], [], []
    for it in range(2):
        loss = ops.train_step(d["st"], fs, buf, U.dev(v[f"map_coord{it}"]), U.dev(v[f"map_label{it}"]), torch.ones(bs, device="cuda"),
                              U.dev(v[f"map_ts{it}"], torch.int32), cert, tsu, gfeat, gdec, sigma=d["sdf_scale"],
                              weight_e=float(v["cfg_weight_e"]), eik_eps=float(v["map_eps"]))
        lab = U.dev(v[f"map_sem{it}"], torch.int32)
        sel, cnt = ops.sem_select(lab, bool(v["cfg_freespace_label_on"]), int(v["cfg_sem_label_decimation"]))
        if it == 0:
            buf.sem_loss = None
        sl = ops.train_sem_step(fsem, buf, lab, sel, cnt, gfeat, gsem, heads=S, weight_s=float(v["cfg_weight_s"]))
        gf, gd, gs = v[f"map_gfeat{it}"], v[f"map_gdec{it}"], v[f"map_gsem{it}"]
        gfs.append(gfeat.cpu().numpy()); gds.append(gdec.cpu().numpy()); gss.append(gsem.cpu().numpy())
        assert np.max(np.abs(gfs[-1] - gf)) < 1e-4 * np.abs(gf).max()
        assert np.max(np.abs(gds[-1] - gd)) < 1e-4 * np.abs(gd).max()
        assert np.max(np.abs(gss[-1] - gs)) < 1e-4 * np.abs(gs).max()
        sem_loss = float(sl.item()) / int(cnt.item())
        sl.zero_()
        assert abs(sem_loss - v["map_loss_sem"][it]) < 1e-4 * abs(v["map_loss_sem"][it])
        l_bce, l_eik = loss.cpu().numpy()
        total = l_bce / bs + float(v["cfg_weight_e"]) * l_eik / n_eik + float(v["cfg_weight_s"]) * sem_loss
        assert abs(total - v["map_loss_total"][it]) < 1e-4 * abs(v["map_loss_total"][it])
        ops.adam_step(feats, gfeat, state[0], state[1], it + 1, lr, eps=aeps)
        ops.adam_step(dec, gdec, state[2], state[3], it + 1, lr, eps=aeps)
        ops.adam_step(sdec, gsem, state[4], state[5], it + 1, lr, eps=aeps)
    G.adam_outliers(feats.cpu().numpy(), v["map_feat_after"], gfs, [v["map_gfeat0"], v["map_gfeat1"]], lr)
    G.adam_outliers(dec.cpu().numpy(), v["map_dec_after"], gds, [v["map_gdec0"], v["map_gdec1"]], lr)
    G.adam_outliers(sdec.cpu().numpy(), v["map_sem_after"], gss, [v["map_gsem0"], v["map_gsem1"]], lr)


def test_semantic_step_any_depth_vs_oracle(sg):
    """Decoder shapes the fixtures do not hold (3 x 64, 4 x 32, 1 x 32, 7 and 32 heads), a frozen semantic decoder, a batch
    that is no multiple of 64: feature / decoder gradients and the loss against the oracle (float64)."""
    from pin_slam_amd import ops, synth
    from tests import gpu_util as U
    d = sg
    k, wf = int(d["query_nn_k"]), bool(d["weighted_first"])
    rng = np.random.default_rng(21)
    coord = d["v"]["map_coord0"][:333]
    for (H, L, S) in ((64, 3, 7), (32, 4, 32), (32, 1, 21)):
        flat = synth.init_decoder(H, L, seed=5 + S, out_dim=S)
        flat[-(S * H + S):] *= 4.0
        lab = rng.integers(-1, S, len(coord)).astype(np.int32)
        feats = U.dev(d["local_geo_features"])
        fsem = dataclasses.replace(d["fs"], feats=feats, dec=U.dev(flat), hidden=H, levels=L, certainty=None, out_dim=S)
        buf = ops.TrainBuffers(len(coord), 10, k, H, L, weighted_first=wf, eikonal=False)
        q = U.dev(coord)
        buf.query.copy_(q)
        ops.knn_query(d["st"], buf.query, k, out=(buf.nbr, buf.nn, None))
        sel, cnt = ops.sem_select(U.dev(lab, torch.int32), False, 2)
        for train_dec in (True, False):
            gfeat, gsem = torch.zeros_like(feats), torch.zeros_like(fsem.dec)
            buf.sem_loss = None
            sl = ops.train_sem_step(fsem, buf, U.dev(lab, torch.int32), sel, cnt, gfeat, gsem if train_dec else None, heads=S, weight_s=0.7)

            def plain(points):
                return O.query_feature(points, _search(d, points), d["local_geo_features"], d["local_neural_points"], None, k,
                                       global2local=d["global2local"], weighted_first=False)

            r = O.train_sem_step(coord, lab, plain, d["local_geo_features"], flat, (11, H, L, S), k, weighted_first=wf, weight_s=0.7,
                                 decimation=2)
            assert int(cnt.item()) == int(r["selected"].sum()) > 0
            assert np.max(np.abs(gfeat.cpu().numpy() - r["feat_grad"])) < 1e-4 * np.abs(r["feat_grad"]).max(), (H, L, S)
            if train_dec:
                assert np.max(np.abs(gsem.cpu().numpy() - r["dec_grad"])) < 1e-4 * np.abs(r["dec_grad"]).max(), (H, L, S)
            else:
                assert float(gsem.abs().max()) == 0.0
            assert abs(float(sl.item()) / int(cnt.item()) - r["loss"]) < 1e-4 * abs(r["loss"])


def test_process_frame_carries_the_semantic_labels():
    """SamplePool(semantic=True): the sampler's labels and the label pool after every frame's window / discard compaction equal
    the reference's sem_label_pool (bit-exact), beside the SDF labels."""
    from pin_slam_amd import pool as P
    from tests.test_gpu_process import _cfg
    d = G.load("process_sem")
    pool = P.SamplePool(semantic=True, capacity=1024)
    for t in range(int(d["n_frames"])):
        f = f"f{t}_"
        scan = torch.from_numpy(d[f + "scan"]).cuda()
        rnd = tuple(torch.from_numpy(d[f + k]).cuda() for k in ("rnd_surface", "rnd_front", "rnd_behind"))
        n0 = pool.n
        pool.append_samples(scan, P.sample_params(_cfg(d), d[f + "pose"], t), rnd=rnd, sem_labels=torch.from_numpy(d[f + "labels"]).cuda())
        A = 1 + int(d["surface_sample_n"]) + int(d["free_front_n"]) + int(d["free_behind_n"])
        assert np.array_equal(pool.view("sem_label")[n0:].cpu().numpy(),
                              O.sample_sem_labels(d[f + "labels"], int(d["surface_sample_n"]), int(d["free_front_n"]), int(d["free_behind_n"])))
        assert pool.n - n0 == A * len(scan)
        disc = torch.from_numpy(d[f + "discard_index"]).cuda()
        n_pool, n_cur = pool.filter(d[f + "pose"][:3, 3], float(d["window_radius"]), int(d["pool_capacity"]),
                                    discard_index=disc if disc.numel() else None)
        assert (n_pool, n_cur) == (int(d[f + "pool_sample_count"]), int(d[f + "cur_sample_count"]))
        assert np.array_equal(pool.view("sem_label").cpu().numpy(), d[f + "after_sem_label_pool"]), t
        assert np.array_equal(pool.view("sdf_label").cpu().numpy().view(np.uint32), d[f + "after_sdf_label_pool"].view(np.uint32)), t


def test_dropin_semantic_slam_loop():
    """The drop-in classes with config.semantic_on (run_demo_sem.yaml's switch): process_frame with per-point labels, mapping with
    the NLL term (the semantic loss falls, the geometry still trains), query_sem through Tracker and Mesher, get_batch's labels."""
    from pin_slam_amd import synth
    from pin_slam_amd.config import PinConfig
    from pin_slam_amd.dropin.model.decoder import Decoder
    from pin_slam_amd.dropin.model.neural_points import NeuralPoints
    from pin_slam_amd.dropin.utils.mapper import Mapper
    from pin_slam_amd.dropin.utils.mesher import Mesher
    from pin_slam_amd.dropin.utils.tracker import Tracker
    from tests.test_gpu_process import _Dataset as _DS
    cfg = PinConfig(voxel_size_m=0.4, buffer_size=int(5e6), local_map_radius=40.0, local_map_travel_dist_ratio=5.0, search_alpha=0.5,
                    query_nn_k=8, bs=4096, bs_new_sample=512, semantic_on=True, sem_class_count=5, pool_filter_freq=1, max_range=40.0)
    torch.manual_seed(3)
    rng = np.random.default_rng(4)
    npts = NeuralPoints(cfg)
    npts.travel_dist = torch.zeros(4, device="cuda")
    dec, sem = Decoder(cfg, 32, 2, 1), Decoder(cfg, 32, 2, cfg.sem_class_count + 1)
    decs = {"sdf": dec, "semantic": sem, "color": None}
    mp = Mapper(cfg, _DS(4), npts, decs)
    pts, _ = synth.disc_points(rng, 20_000, 18.0, 2)
    # labels by quadrant of the (x, y) plane + the sheet: a field the head can learn
    lab = (1 + (pts[:, 0] > 0).astype(np.int32) + 2 * (pts[:, 1] > 0).astype(np.int32)).astype(np.int32)
    pose = torch.eye(4, dtype=torch.float64, device="cuda")
    mp.process_frame(torch.from_numpy(pts).cuda(), torch.from_numpy(lab).cuda(), pose, 0)
    assert mp.sem_label_pool is not None and mp.sem_label_pool.shape[0] == mp.pool_sample_count
    pool_lab = mp.sem_label_pool.cpu().numpy()
    assert set(np.unique(pool_lab)) <= {0, 1, 2, 3, 4}
    b = mp.get_batch()
    assert b[4] is not None and b[4].dtype == torch.int32 and b[4].shape[0] == cfg.bs
    t = mp._get_trainer()
    losses = []
    for _ in range(6):
        mp.mapping(10)
        losses.append(float(t.sem_loss.item()) / max(int(t.sem_count.item()), 1))
        t.sem_loss.zero_()
    assert losses[-1] < 0.6 * losses[0], losses  # (per call: sum over its 10 iterations / the last count)
    trk = Tracker(cfg, npts, decs)
    q = torch.from_numpy(pts[:4000]).cuda()
    res = trk.query_source_points(q, cfg.infer_bs, True, False, False, False, query_sem=True)
    pred = res[4].cpu().numpy().astype(np.int32)
    assert (pred == lab[:4000]).mean() > 0.8, (pred == lab[:4000]).mean()
    ms = Mesher(cfg, npts, decs)
    _, sem_pred, _, mask = ms.query_points(q, 4096, True, True, False, True, query_locally=True, out_torch=True)
    assert np.array_equal(sem_pred.numpy().astype(np.int32), pred)
    lp = sem.sem_label_prob(torch.randn(10, 11, device="cuda"))
    assert lp.shape == (10, cfg.sem_class_count + 1) and torch.allclose(lp.exp().sum(-1), torch.ones(10, device="cuda"), atol=1e-5)
