"""Generate tests/golden/*.npz by running the UNMODIFIED reference on CPU.

TEST INFRASTRUCTURE.  Run inside the build container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden

Each fixture holds the complete inputs of a hot-path call (map arrays, sparse hash
table, decoder parameters, query points, config scalars) and the reference's
outputs, so the tests never need the reference tree again.  The reference has no
golden vectors of its own -- these are the pins.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

from . import ref_loader as R

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

CASES = {
    # name: (config overrides, decoder (levels, hidden))
    "c2_wf": (dict(voxel_size_m=0.4, search_alpha=0.5, query_nn_k=8, weighted_first=True,
                   buffer_size=40009), (2, 32)),
    "kitti_nwf": (dict(voxel_size_m=0.4, search_alpha=0.2, query_nn_k=6, weighted_first=False,
                       buffer_size=40009), (1, 64)),
    "c3_bigtable": (dict(voxel_size_m=0.4, search_alpha=0.5, query_nn_k=8, weighted_first=True,
                         buffer_size=int(5e7)), (4, 64)),
}


def sheet_points(gen, n, radius, center=(0.0, 0.0), layers=2):
    r = radius * torch.sqrt(torch.rand(n, generator=gen))
    th = 2 * np.pi * torch.rand(n, generator=gen)
    x = center[0] + r * torch.cos(th)
    y = center[1] + r * torch.sin(th)
    l = torch.randint(0, layers, (n,), generator=gen).float()
    z = -2 + 0.8 * l + 0.3 * torch.sin(0.5 * x) * torch.cos(0.5 * y)
    return torch.stack([x, y, z], 1).float()


def flat_decoder(dec):
    return torch.cat([p.detach().reshape(-1) for p in dec.state_dict().values()]).numpy().copy()


def t2n(t):
    return t.detach().cpu().numpy().copy()


class LossSpy:
    """Record the scalar losses of every Mapper.mapping iteration of the reference: the BCE term returned by
    utils.loss.sdf_bce_loss (mapper.py:733) and the total `cur_loss` at its backward() call (mapper.py:817)."""

    def __init__(self, mapper_mod):
        self.mod, self.sdf, self.total = mapper_mod, [], []

    def __enter__(self):
        self._bce, self._bwd = self.mod.sdf_bce_loss, torch.Tensor.backward

        def bce(*a, **k):
            v = self._bce(*a, **k)
            self.sdf.append(float(v.detach()))
            return v

        def backward(t, *a, **k):
            if t.dim() == 0:
                self.total.append(float(t.detach()))
            return self._bwd(t, *a, **k)

        self.mod.sdf_bce_loss, torch.Tensor.backward = bce, backward
        return self

    def __exit__(self, *exc):
        self.mod.sdf_bce_loss, torch.Tensor.backward = self._bce, self._bwd
        return False


def build(case):
    m = R.load()
    over, (levels, hidden) = CASES[case]
    cfg = R.make_config(local_map_radius=20.0, local_map_travel_dist_ratio=1.0, bs=512,
                        feature_std=0.1, gradient_decimation=10, track_on=True, **over)
    cfg.geo_mlp_level, cfg.geo_mlp_hidden_dim = levels, hidden
    torch.manual_seed(42)
    dec = m["Decoder"](cfg, hidden, levels, 1)
    npts = m["NeuralPoints"](cfg)
    gen = torch.Generator().manual_seed(0)
    # three frames, sensor moving 8 m per frame along x: the first frame ends up outside
    # the travel-distance window (20 m * 1.0) of the last one only partially -> the time
    # filter and the local-map mask both bite.
    travel = [0.0, 12.0, 24.0]
    npts.travel_dist = torch.tensor(travel, dtype=torch.float32)
    for ts in range(3):
        pts = sheet_points(gen, 20000, 14.0, center=(8.0 * ts, 0.0))
        npts.update(pts, torch.tensor([8.0 * ts, 0.0, 0.0]), torch.eye(3), ts)
    # non-trivial certainties / features (update draws features with torch.randn)
    npts.point_certainties = torch.rand(npts.count(), generator=gen) * 3.0
    npts.reset_local_map(torch.tensor([16.0, 0.0, 0.0]), torch.eye(3), 2)
    pretrain(m, cfg, dec, npts, gen)
    return m, cfg, dec, npts, gen


def surface_samples(gen, n, radius, center, sigma=0.12):
    """Points on the analytic sheets displaced by d along the surface normal; label = d."""
    base = sheet_points(gen, n, radius, center=center)
    x, y = base[:, 0], base[:, 1]
    fx = 0.15 * torch.cos(0.5 * x) * torch.cos(0.5 * y)
    fy = -0.15 * torch.sin(0.5 * x) * torch.sin(0.5 * y)
    nrm = torch.stack([-fx, -fy, torch.ones_like(fx)], 1)
    nrm = nrm / nrm.norm(dim=1, keepdim=True)
    d = sigma * torch.randn(n, generator=gen)
    return (base + d[:, None] * nrm).float(), d.float()


def pretrain(m, cfg, dec, npts, gen, iters=150):
    """Train features + decoder with the reference's own Mapper.mapping so that the SDF is a
    meaningful field (tracking converges, gradients have norm ~1)."""
    ds = R.FakeDataset(n_frames=3)
    mp = m["Mapper"](cfg, ds, npts, {"sdf": dec, "semantic": None, "color": None})
    mp.determine_used_pose()

    def get_batch(global_coord=False):
        c, l = surface_samples(gen, cfg.bs, 13.0, (16.0, 0.0))
        ts = torch.full((cfg.bs,), 2, dtype=torch.int)
        return c, l, ts, None, None, None, torch.ones(cfg.bs)

    mp.get_batch = get_batch
    mp.mapping(iters)


def map_arrays(npts, cfg, dec):
    tab = npts.buffer_pt_index
    slots = torch.nonzero(tab >= 0).flatten()
    d = dict(
        buffer_size=np.int64(cfg.buffer_size), table_slots=t2n(slots), table_vals=t2n(tab[slots]),
        neural_points=t2n(npts.neural_points), point_orientations=t2n(npts.point_orientations),
        geo_features=t2n(npts.geo_features), point_ts_create=t2n(npts.point_ts_create),
        point_ts_update=t2n(npts.point_ts_update), point_certainties=t2n(npts.point_certainties),
        local_mask=t2n(npts.local_mask), global2local=t2n(npts.global2local),
        local_neural_points=t2n(npts.local_neural_points),
        local_geo_features=t2n(npts.local_geo_features.data),
        local_point_certainties=t2n(npts.local_point_certainties),
        local_point_ts_update=t2n(npts.local_point_ts_update),
        local_point_orientations=t2n(npts.local_point_orientations),
        travel_dist=t2n(npts.travel_dist), cur_ts=np.int64(npts.cur_ts),
        diff_travel_dist_local=np.float64(npts.diff_travel_dist_local),
        local_map_radius=np.float64(npts.local_map_radius),
        resolution=np.float64(npts.resolution), neighbor_dx=t2n(npts.neighbor_dx),
        max_valid_dist2=np.float64(npts.max_valid_dist2), query_nn_k=np.int64(cfg.query_nn_k),
        weighted_first=np.bool_(cfg.weighted_first), num_nei_cells=np.int64(cfg.num_nei_cells),
        search_alpha=np.float64(cfg.search_alpha),
        dec_flat=flat_decoder(dec), dec_levels=np.int64(len(dec.layers)),
        dec_hidden=np.int64(dec.layers[0].out_features), sdf_scale=np.float64(dec.sdf_scale),
    )
    return d


def mapping_section(m, cfg, dec, npts, gen, out, iters=2, tag="map"):
    """Mapper.mapping on FIXED batches (get_batch RNG bypassed): inputs, per-iteration autograd gradients and scalar
    losses, parameters and side effects afterwards, under the keys `tag`_*."""
    ds = R.FakeDataset(n_frames=3)
    mp = m["Mapper"](cfg, ds, npts, {"sdf": dec, "semantic": None, "color": None})
    mp.determine_used_pose()
    bs = cfg.bs
    batches = []
    for it in range(iters):
        coord, label = surface_samples(gen, bs, 12.0, (16.0, 0.0), sigma=0.2)
        label = (label + 0.02 * torch.randn(bs, generator=gen)).float()
        ts = torch.randint(0, 3, (bs,), generator=gen).int()
        w = (0.6 + 0.8 * torch.rand(bs, generator=gen)).float() * torch.where(
            torch.rand(bs, generator=gen) < 0.5, 1.0, -1.0)
        batches.append((coord, label, ts, w))
        out[f"{tag}_coord{it}"] = t2n(coord); out[f"{tag}_label{it}"] = t2n(label)
        out[f"{tag}_ts{it}"] = t2n(ts); out[f"{tag}_w{it}"] = t2n(w)
    it_box = {"i": 0}

    def fake_get_batch(global_coord=False):
        c, l, t, w = batches[it_box["i"]]
        it_box["i"] += 1
        return c.clone(), l.clone(), t.clone(), None, None, None, w.clone()

    mp.get_batch = fake_get_batch
    grads = []
    real_setup = m["mapper_mod"].setup_optimizer

    def spy_setup(*a, **k):
        opt = real_setup(*a, **k)
        real_step = opt.step

        def step(*aa, **kk):
            g = {}
            g["feat"] = t2n(npts.local_geo_features.grad)
            g["dec"] = np.concatenate([t2n(p.grad).ravel() for p in dec.parameters()])
            grads.append(g)
            return real_step(*aa, **kk)

        opt.step = step
        return opt

    m["mapper_mod"].setup_optimizer = spy_setup
    out[f"{tag}_eps"] = np.float64(cfg.voxel_size_m * cfg.num_grad_step_ratio)
    out[f"{tag}_dec"] = np.int64(cfg.gradient_decimation)
    out[f"{tag}_weight_e"] = np.float64(cfg.weight_e)
    out[f"{tag}_lr"] = np.float64(cfg.lr); out[f"{tag}_adam_eps"] = np.float64(cfg.adam_eps)
    out[f"{tag}_loss_weight_on"] = np.bool_(cfg.loss_weight_on)
    try:
        with LossSpy(m["mapper_mod"]) as spy:
            mp.mapping(iters)
    finally:
        m["mapper_mod"].setup_optimizer = real_setup
    out[f"{tag}_loss_sdf"] = np.asarray(spy.sdf, np.float64); out[f"{tag}_loss_total"] = np.asarray(spy.total, np.float64)
    for it, g in enumerate(grads):
        out[f"{tag}_gfeat{it}"] = g["feat"]; out[f"{tag}_gdec{it}"] = g["dec"]
    out[f"{tag}_feat_after"] = t2n(npts.local_geo_features.data)
    out[f"{tag}_dec_after"] = flat_decoder(dec)
    out[f"{tag}_cert_after"] = t2n(npts.local_point_certainties)
    out[f"{tag}_ts_after"] = t2n(npts.local_point_ts_update)
    out[f"{tag}_global_feat_after"] = t2n(npts.geo_features)


def gen_case(case):
    m, cfg, dec, npts, gen = build(case)
    out = map_arrays(npts, cfg, dec)
    tools = m["tools"]

    # ---- queries: points near the sheets, some far (no neighbours), some exactly on voxel faces
    q = sheet_points(gen, 400, 13.0, center=(16.0, 0.0)) + 0.05 * torch.randn(400, 3, generator=gen)
    far = torch.tensor([[200.0, 200.0, 50.0], [-300.0, 10.0, 0.0]])
    face = torch.floor(q[:30] / cfg.voxel_size_m) * cfg.voxel_size_m  # on cell boundaries
    q = torch.cat([q, far, face], 0).float().contiguous()
    out["query"] = t2n(q)

    # (1) radius_neighborhood_search with and without the travel-distance filter
    for tf in (False, True):
        d2, idx = npts.radius_neighborhood_search(q.clone(), time_filtering=tf)
        out[f"rs_d2_tf{int(tf)}"] = t2n(d2)
        out[f"rs_idx_tf{int(tf)}"] = t2n(idx)

    # (2) query_feature, inference mode; local and global; then after_pgo (rotated vectors)
    for tag, loc in (("loc", True), ("glob", False)):
        gf, _, w, nn, cert = npts.query_feature(q.clone(), training_mode=False, query_locally=loc)
        out[f"qf_{tag}_feat"] = t2n(gf); out[f"qf_{tag}_w"] = t2n(w)
        out[f"qf_{tag}_nn"] = t2n(nn); out[f"qf_{tag}_cert"] = t2n(cert)

    # (3,4) Tracker.query_source_points: sdf, autograd gradient, mask, certainty, std
    trk = m["Tracker"](cfg, npts, {"sdf": dec, "semantic": None, "color": None})
    res = trk.query_source_points(q.clone(), cfg.infer_bs, True, True, False, False,
                                  query_locally=True, mask_min_nn_count=cfg.track_mask_query_nn_k)
    sdf, grad, _, _, _, mask, cert, std = res
    out["qsp_sdf"] = t2n(sdf); out["qsp_grad"] = t2n(grad); out["qsp_mask"] = t2n(mask)
    out["qsp_cert"] = t2n(cert); out["qsp_std"] = t2n(std)
    out["track_mask_query_nn_k"] = np.int64(cfg.track_mask_query_nn_k)

    # (4b) after_pgo: neighbour vectors rotated by per-point quaternions (neural_points.py:645-648)
    quat = torch.randn(npts.local_point_orientations.shape, generator=gen)
    quat = (quat / quat.norm(dim=1, keepdim=True)).float()
    saved_q = npts.local_point_orientations
    npts.local_point_orientations, npts.after_pgo = quat, True
    res = trk.query_source_points(q.clone(), cfg.infer_bs, True, True, False, False,
                                  query_locally=True, mask_min_nn_count=cfg.track_mask_query_nn_k)
    out["pgo_quat"] = t2n(quat); out["pgo_sdf"] = t2n(res[0]); out["pgo_grad"] = t2n(res[1])
    out["pgo_std"] = t2n(res[7])
    gf, _, w, nn, cert = npts.query_feature(q.clone(), training_mode=False, query_locally=True)
    out["pgo_feat"] = t2n(gf)
    npts.local_point_orientations, npts.after_pgo = saved_q, False

    # (5) one registration step + full tracking on a bigger, slightly misaligned scan
    src = sheet_points(gen, 3000, 12.0, center=(16.0, 0.0), layers=2)
    Tinit = torch.eye(4, dtype=torch.float64)
    ang = 0.01
    Tinit[:3, :3] = torch.tensor([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]])
    Tinit[:3, 3] = torch.tensor([0.05, -0.03, 0.02], dtype=torch.float64)
    out["reg_src"] = t2n(src); out["reg_Tinit"] = t2n(Tinit)
    cur = tools.transform_torch(src, Tinit)
    out["reg_cur"] = t2n(cur)
    step = trk.registration_step(cur.clone(), None, torch.zeros(len(src)), None,
                                 cfg.reg_min_grad_norm, cfg.reg_max_grad_norm,
                                 cfg.reg_GM_dist_m, cfg.reg_GM_grad, cfg.reg_lm_lambda, False)
    out["reg_dT"] = t2n(step[0]); out["reg_valid_count"] = np.int64(step[4].shape[0])
    out["reg_residual_cm"] = np.float64(step[5])
    cfg.reg_iter_n = 20
    Tfin, cov, _, vflag = trk.tracking(src.clone(), Tinit.clone())
    out["trk_T"] = t2n(Tfin); out["trk_valid"] = np.bool_(vflag)
    out["trk_cov"] = np.asarray(cov) if cov is not None else np.zeros((0,))
    for kname in ("reg_min_grad_norm", "reg_max_grad_norm", "reg_GM_dist_m", "reg_GM_grad",
                  "reg_lm_lambda", "surface_sample_range_m", "max_sdf_std_ratio", "reg_iter_n",
                  "reg_term_thre_deg", "reg_term_thre_m", "final_residual_ratio_thre",
                  "eigenvalue_ratio_thre"):
        out["cfg_" + kname] = np.float64(getattr(cfg, kname))

    # (6) Mapper.mapping: two iterations on FIXED batches
    mapping_section(m, cfg, dec, npts, gen, out)
    return out


# config/lidar_slam/run_livox.yaml: numerical_grad_on False (analytic Eikonal on every sample), weighted_first False, k = 8
ANALYTIC = (dict(voxel_size_m=0.4, search_alpha=0.5, query_nn_k=8, weighted_first=False, buffer_size=40009), (1, 64))


def gen_analytic():
    """Mapper.mapping with the Eikonal term on the AUTOGRAD gradient (mapper.py:642-643, 677-678 with
    numerical_grad False, config.py:437-439): the map of the other cases, then three short runs from recorded
    states -- per-neighbour decoding (run_livox.yaml), the same after a pose-graph correction (neighbour vectors
    rotated by the point orientations), and weighted-first decoding."""
    CASES["analytic_eik"] = ANALYTIC
    try:
        m, cfg, dec, npts, gen = build("analytic_eik")
    finally:
        del CASES["analytic_eik"]
    cfg.numerical_grad, cfg.gradient_decimation, cfg.weight_e = False, 1, 0.5
    out = map_arrays(npts, cfg, dec)
    for tag, iters in (("nwf", 2), ("pgo", 1), ("wf", 1)):
        if tag == "pgo":
            quat = torch.randn(npts.local_point_orientations.shape, generator=gen)
            npts.local_point_orientations, npts.after_pgo = (quat / quat.norm(dim=1, keepdim=True)).float(), True
            out["pgo_quat"] = t2n(npts.local_point_orientations)
        cfg.weighted_first = tag == "wf"
        out[f"{tag}_feat_before"] = t2n(npts.local_geo_features.data)
        out[f"{tag}_dec_before"] = flat_decoder(dec)
        out[f"{tag}_cert_before"] = t2n(npts.local_point_certainties)
        out[f"{tag}_tsu_before"] = t2n(npts.local_point_ts_update)
        mapping_section(m, cfg, dec, npts, gen, out, iters=iters, tag=tag)
        for key in ("global_feat_after", "feat_after", "dec_after"):  # (the next run's *_before; not compared)
            del out[f"{tag}_{key}"]
        if tag == "pgo":
            npts.after_pgo = False
    return out


def color_of(p):
    """Smooth synthetic colour field in [0.1, 0.9]^3."""
    x, y = p[:, 0], p[:, 1]
    return (0.5 + 0.4 * torch.stack([torch.sin(0.7 * x), torch.cos(0.5 * y), torch.sin(0.3 * x + 0.4 * y)], 1)).float()


def gen_color_case():
    """C5-style colour path (run_replica.yaml: colour features + colour decoder + photometric
    registration): reference outputs for query_feature(colour), regress_color + per-channel
    autograd gradients, registration with the photometric term / the consistency weight, and
    two mapping iterations with the colour loss."""
    m = R.load()
    cfg = R.make_config(local_map_radius=20.0, local_map_travel_dist_ratio=1.0, bs=512, feature_std=0.1,
                        gradient_decimation=10, track_on=True, voxel_size_m=0.4, search_alpha=0.2, query_nn_k=6,
                        weighted_first=True, buffer_size=int(5e7), color_on=True, color_channel=3,
                        photometric_loss_on=True, photometric_loss_weight=0.01, consist_wieght_on=True, weight_i=1.0)
    torch.manual_seed(7)
    dec = m["Decoder"](cfg, 64, 1, 1)
    cdec = m["Decoder"](cfg, 64, 1, 3)
    decoders = {"sdf": dec, "semantic": None, "color": cdec}
    npts = m["NeuralPoints"](cfg)
    gen = torch.Generator().manual_seed(5)
    npts.travel_dist = torch.tensor([0.0, 12.0, 24.0], dtype=torch.float32)
    for ts in range(3):
        pts = sheet_points(gen, 20000, 14.0, center=(8.0 * ts, 0.0))
        npts.update(pts, torch.tensor([8.0 * ts, 0.0, 0.0]), torch.eye(3), ts)
    npts.point_certainties = torch.rand(npts.count(), generator=gen) * 3.0
    npts.reset_local_map(torch.tensor([16.0, 0.0, 0.0]), torch.eye(3), 2)
    ds = R.FakeDataset(n_frames=3)
    mp = m["Mapper"](cfg, ds, npts, decoders)
    mp.determine_used_pose()

    def train_batch(global_coord=False):
        c, l = surface_samples(gen, cfg.bs, 13.0, (16.0, 0.0))
        ts = torch.full((cfg.bs,), 2, dtype=torch.int)
        return c, l, ts, None, None, color_of(c), torch.ones(cfg.bs)

    mp.get_batch = train_batch
    mp.mapping(200)
    out = map_arrays(npts, cfg, dec)
    out.update(color_features=t2n(npts.color_features), local_color_features=t2n(npts.local_color_features.data),
               cdec_flat=flat_decoder(cdec), cdec_levels=np.int64(1), cdec_hidden=np.int64(64),
               surface_sample_range_m=np.float64(cfg.surface_sample_range_m), weight_i=np.float64(cfg.weight_i),
               photometric_loss_weight=np.float64(cfg.photometric_loss_weight))
    q = sheet_points(gen, 300, 13.0, center=(16.0, 0.0)) + 0.05 * torch.randn(300, 3, generator=gen)
    q = q.float().contiguous()
    out["query"] = t2n(q)
    gf, cf, w, nn, cert = npts.query_feature(q.clone(), training_mode=False, query_locally=True, query_color_feature=True)
    out["qf_color_feat"] = t2n(cf); out["qf_geo_feat"] = t2n(gf)
    trk = m["Tracker"](cfg, npts, decoders)
    res = trk.query_source_points(q.clone(), cfg.infer_bs, True, True, True, True, query_locally=True,
                                  mask_min_nn_count=cfg.track_mask_query_nn_k)
    out["qsp_sdf"] = t2n(res[0]); out["qsp_grad"] = t2n(res[1]); out["qsp_color"] = t2n(res[2])
    out["qsp_color_grad"] = t2n(res[3]); out["qsp_mask"] = t2n(res[5])
    out["track_mask_query_nn_k"] = np.int64(cfg.track_mask_query_nn_k)
    # registration with colours: photometric term on, then off (consistency weight instead)
    src = sheet_points(gen, 3000, 12.0, center=(16.0, 0.0), layers=2)
    Tinit = torch.eye(4, dtype=torch.float64)
    ang = 0.008
    Tinit[:3, :3] = torch.tensor([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]])
    Tinit[:3, 3] = torch.tensor([0.04, -0.03, 0.02], dtype=torch.float64)
    tools = m["tools"]
    cur = tools.transform_torch(src, Tinit)
    src_col = (color_of(src) + 0.02 * torch.randn(len(src), 3, generator=gen)).clamp(0, 1).float()
    out["reg_src"] = t2n(src); out["reg_Tinit"] = t2n(Tinit); out["reg_cur"] = t2n(cur); out["reg_colors"] = t2n(src_col)
    for tag, photo in (("photo", True), ("consist", False)):
        cfg.photometric_loss_on = photo
        step = trk.registration_step(cur.clone(), None, torch.zeros(len(src)), src_col.clone(),
                                     cfg.reg_min_grad_norm, cfg.reg_max_grad_norm, cfg.reg_GM_dist_m, cfg.reg_GM_grad,
                                     cfg.reg_lm_lambda, False)
        out[f"reg_dT_{tag}"] = t2n(step[0]); out[f"reg_valid_{tag}"] = np.int64(step[4].shape[0])
        out[f"reg_res_{tag}"] = np.float64(step[5])
        out[f"reg_photo_res_{tag}"] = np.float64(step[6] if step[6] is not None else -1.0)
    cfg.photometric_loss_on = True
    cfg.reg_iter_n = 20
    Tfin, cov, _, vflag = trk.tracking(src.clone(), Tinit.clone(), source_colors=src_col.clone())
    out["trk_T"] = t2n(Tfin); out["trk_valid"] = np.bool_(vflag)
    for kname in ("reg_min_grad_norm", "reg_max_grad_norm", "reg_GM_dist_m", "reg_GM_grad", "reg_lm_lambda",
                  "max_sdf_std_ratio", "reg_iter_n", "reg_term_thre_deg", "reg_term_thre_m"):
        out["cfg_" + kname] = np.float64(getattr(cfg, kname))
    # two mapping iterations on fixed batches with colour labels
    bs = cfg.bs
    batches = []
    for it in range(2):
        coord, label = surface_samples(gen, bs, 12.0, (16.0, 0.0), sigma=0.2)
        ts = torch.randint(0, 3, (bs,), generator=gen).int()
        w = (0.6 + 0.8 * torch.rand(bs, generator=gen)).float()
        col = (color_of(coord) + 0.05 * torch.randn(bs, 3, generator=gen)).clamp(0, 1).float()
        batches.append((coord, label.float(), ts, w, col))
        for nme, v in zip(("coord", "label", "ts", "w", "color"), batches[-1]):
            out[f"map_{nme}{it}"] = t2n(v)
    box = {"i": 0}

    def fixed_batch(global_coord=False):
        c, l, t, w, col = batches[box["i"]]
        box["i"] += 1
        return c.clone(), l.clone(), t.clone(), None, None, col.clone(), w.clone()

    mp.get_batch = fixed_batch
    grads = []
    real_setup = m["mapper_mod"].setup_optimizer

    def spy_setup(*a, **k):
        opt = real_setup(*a, **k)
        real_step = opt.step

        def step(*aa, **kk):
            grads.append(dict(gfeat=t2n(npts.local_geo_features.grad), cfeat=t2n(npts.local_color_features.grad),
                              gdec=np.concatenate([t2n(p.grad).ravel() for p in dec.parameters()]),
                              cdec=np.concatenate([t2n(p.grad).ravel() for p in cdec.parameters()])))
            return real_step(*aa, **kk)

        opt.step = step
        return opt

    m["mapper_mod"].setup_optimizer = spy_setup
    out["map_eps"] = np.float64(cfg.voxel_size_m * cfg.num_grad_step_ratio)
    out["map_dec"] = np.int64(cfg.gradient_decimation); out["map_weight_e"] = np.float64(cfg.weight_e)
    out["map_lr"] = np.float64(cfg.lr); out["map_adam_eps"] = np.float64(cfg.adam_eps)
    try:
        with LossSpy(m["mapper_mod"]) as spy:
            mp.mapping(2)
    finally:
        m["mapper_mod"].setup_optimizer = real_setup
    out["map_loss_sdf"] = np.asarray(spy.sdf, np.float64); out["map_loss_total"] = np.asarray(spy.total, np.float64)
    for it, g in enumerate(grads):
        for kk, v in g.items():
            out[f"map_{kk}{it}"] = v
    out["map_geo_after"] = t2n(npts.local_geo_features.data); out["map_color_after"] = t2n(npts.local_color_features.data)
    out["map_gdec_after"] = flat_decoder(dec); out["map_cdec_after"] = flat_decoder(cdec)
    return out


def gen_update():
    """(7) NeuralPoints.update / reset_local_map resulting arrays (K8/K9 parity)."""
    m = R.load()
    cfg = R.make_config(voxel_size_m=0.4, buffer_size=int(5e7), local_map_radius=20.0,
                        local_map_travel_dist_ratio=1.0)
    npts = m["NeuralPoints"](cfg)
    gen = torch.Generator().manual_seed(3)
    out = {}
    travel = [0.0, 9.0, 18.0, 27.0]
    npts.travel_dist = torch.tensor(travel, dtype=torch.float32)
    out["travel_dist"] = np.asarray(travel, np.float32)
    for ts in range(4):
        pts = sheet_points(gen, 8000, 12.0, center=(9.0 * ts, 0.0))
        out[f"pts{ts}"] = t2n(pts)
        out[f"sel{ts}"] = t2n(m["tools"].voxel_down_sample_torch(pts, cfg.voxel_size_m))
        npts.update(pts, torch.tensor([9.0 * ts, 0.0, 0.0]), torch.eye(3), ts)
        out[f"count{ts}"] = np.int64(npts.count())
        out[f"local_mask{ts}"] = t2n(npts.local_mask)
        out[f"global2local{ts}"] = t2n(npts.global2local)
    # reset_local_map variants on the final map: a window of FRAMES instead of travel distance (pin_slam.py:287 passes
    # config.loop_local_map_by_travel_dist = False), a float64 sensor position (dataset.cur_pose_torch[:3, 3]: the radius
    # test promotes to float64), use_mid_ts
    npts.point_ts_update = torch.randint(0, 4, (npts.count(),), generator=gen).int().maximum(npts.point_ts_create)
    out["point_ts_update"] = t2n(npts.point_ts_update)
    sp64 = torch.tensor([13.37, 0.21, -0.4], dtype=torch.float64)
    out["var_sensor"] = t2n(sp64)
    npts.reset_local_map(sp64.float(), None, 2, False, 2)
    out["var_ts_mask"], out["var_ts_g2l"] = t2n(npts.local_mask), t2n(npts.global2local)
    npts.reset_local_map(sp64, None, 3, True)
    out["var_f64_mask"] = t2n(npts.local_mask)
    npts.reset_local_map(sp64.float(), None, 3, True)
    out["var_f32_mask"] = t2n(npts.local_mask)
    cfg.use_mid_ts = True
    npts.reset_local_map(sp64.float(), None, 3, True)
    out["var_mid_mask"] = t2n(npts.local_mask)
    npts.reset_local_map(sp64.float(), None, 2, False, 1)
    out["var_mid_ts_mask"] = t2n(npts.local_mask)
    cfg.use_mid_ts = False
    tab = npts.buffer_pt_index
    slots = torch.nonzero(tab >= 0).flatten()
    out.update(table_slots=t2n(slots), table_vals=t2n(tab[slots]), neural_points=t2n(npts.neural_points),
               point_ts_create=t2n(npts.point_ts_create), buffer_size=np.int64(cfg.buffer_size),
               resolution=np.float64(cfg.voxel_size_m), local_map_radius=np.float64(npts.local_map_radius),
               diff_travel_dist_local=np.float64(npts.diff_travel_dist_local))
    return out


class _RngSpy:
    """Records every torch.randn / rand / randint result drawn while active (the reference draws
    its sampling noise inside process_frame; the kernels take the same numbers as inputs)."""

    def __init__(self):
        self.calls = []

    def __enter__(self):
        self._orig = {n: getattr(torch, n) for n in ("randn", "rand", "randint")}
        for n, f in self._orig.items():
            def wrap(*a, _f=f, _n=n, **k):
                r = _f(*a, **k)
                self.calls.append((_n, r.detach().clone()))
                return r
            setattr(torch, n, wrap)
        return self

    def __exit__(self, *exc):
        for n, f in self._orig.items():
            setattr(torch, n, f)


def gen_process(color=False):
    """(8) Mapper.process_frame data path (utils/mapper.py:162-449): DataSampler.sample, pool
    append, distance window + random discard, query_certainty and the new-sample index, for four
    frames of a moving sensor with a short Mapper.mapping in between (so certainties are real).
    Config chosen so that every branch bites: window radius smaller than the travelled distance,
    pool capacity exceeded from frame 2 on."""
    m = R.load()
    cfg = R.make_config(voxel_size_m=0.4, buffer_size=40009, local_map_radius=22.0, local_map_travel_dist_ratio=5.0,
                        bs=1024, bs_new_sample=256, feature_std=0.05, track_on=True, pool_capacity=(14000 if color else 33000),
                        pool_filter_freq=1, new_certainty_thre=1.0, surface_sample_range_m=0.25,
                        free_sample_end_dist_m=1.0, max_range=20.0, behind_dropoff_on=bool(color),
                        adaptive_iters=True, search_alpha=0.5, query_nn_k=6)
    cfg.window_radius = 13.0
    if color:
        cfg.color_on, cfg.color_channel = True, 3
    torch.manual_seed(7)
    dec = m["Decoder"](cfg, 32, 1, 1)
    cdec = m["Decoder"](cfg, 32, 1, 3) if color else None
    npts = m["NeuralPoints"](cfg)
    nfr = 3 if color else 4
    ds = R.FakeDataset(n_frames=nfr)
    mp = m["Mapper"](cfg, ds, npts, {"sdf": dec, "semantic": None, "color": cdec})
    gen = torch.Generator().manual_seed(11)
    out = dict(n_frames=np.int64(nfr), buffer_size=np.int64(cfg.buffer_size), resolution=np.float64(cfg.voxel_size_m))
    for k in ("surface_sample_range_m", "surface_sample_n", "free_front_n", "free_behind_n", "free_sample_begin_ratio",
              "free_sample_end_dist_m", "dist_weight_on", "dist_weight_scale", "max_range", "behind_dropoff_on",
              "window_radius", "pool_capacity", "new_certainty_thre", "map_surface_ratio", "bs_new_sample",
              "new_sample_ratio_less", "new_sample_ratio_more", "new_sample_ratio_restart", "freeze_after_frame"):
        out[k] = np.asarray(getattr(cfg, k))
    travel = [0.0]
    for ts in range(nfr):
        a = 0.15 * ts
        pose = np.eye(4)
        pose[:3, :3] = [[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]
        pose[:3, 3] = [6.0 * ts, 0.5 * ts, 0.1 * ts]
        if ts:
            travel.append(travel[-1] + float(np.linalg.norm(pose[:3, 3] - prev[:3, 3])))
        prev = pose
        ds.odom_poses[ts] = pose
        ds.processed_frame = ts
        npts.travel_dist = torch.tensor(travel + [0.0] * (nfr - len(travel)), dtype=torch.float32)
        scan = sheet_points(gen, 1500 if color else 2500, 12.0)
        if color:
            scan = torch.cat([scan, color_of(scan)], 1)
        pool_before = {k: t2n(getattr(mp, k)) for k in ("coord_pool", "global_coord_pool", "sdf_label_pool",
                                                        "weight_pool", "time_pool")}
        if color:
            pool_before["color_pool"] = t2n(mp.color_pool)
        spy_cert = {}
        orig_qc = npts.query_certainty

        def qc(points, _o=orig_qc, _s=spy_cert):
            tab = npts.buffer_pt_index
            slots = torch.nonzero(tab >= 0).flatten()
            r = _o(points)
            _s.update(qc_out=t2n(r), qc_table_slots=t2n(slots), qc_table_vals=t2n(tab[slots]),
                      qc_positions=t2n(npts.neural_points), qc_certainties=t2n(npts.point_certainties))
            return r

        npts.query_certainty = qc
        spy_s = {}
        orig_sample = mp.sampler.sample

        def sample(*a, _o=orig_sample, _s=spy_s):
            r = _o(*a)
            _s["out"] = r
            return r

        mp.sampler.sample = sample
        pose_t = torch.tensor(pose, dtype=torch.float64)
        with _RngSpy() as spy:
            mp.process_frame(scan, None, pose_t, ts)
        mp.sampler.sample, npts.query_certainty = orig_sample, orig_qc
        names = [c[0] for c in spy.calls]
        # draws: randn (surface), rand (front), rand (behind), randn (new point features: geo [, colour]), [randint discard]
        assert names[:3] == ["randn", "rand", "rand"], names
        f = f"f{ts}_"
        out[f + "scan"] = t2n(scan)
        out[f + "pose"] = pose
        out[f + "rnd_surface"], out[f + "rnd_front"], out[f + "rnd_behind"] = (t2n(c[1]).reshape(-1) for c in spy.calls[:3])
        disc = [c[1] for c in spy.calls if c[0] == "randint"]
        out[f + "discard_index"] = t2n(disc[0]) if disc else np.zeros((0,), np.int64)
        co, lab, _, _, col, w = spy_s["out"]
        out[f + "s_coord"], out[f + "s_label"], out[f + "s_weight"] = t2n(co), t2n(lab), t2n(w)
        if color:
            out[f + "s_color"] = t2n(col)
        if ts == 0:
            assert all(v.shape[0] == 0 for v in pool_before.values())  # frame t starts from frame t-1's "after" pools
        for k in pool_before:
            out[f + "after_" + k] = t2n(getattr(mp, k))
        out[f + "cur_sample_count"] = np.int64(mp.cur_sample_count)
        out[f + "pool_sample_count"] = np.int64(mp.pool_sample_count)
        out[f + "new_idx"] = t2n(mp.new_idx)
        out[f + "adaptive_iter_offset"] = np.int64(mp.adaptive_iter_offset)
        out[f + "count"] = np.int64(npts.count())
        for k, v in spy_cert.items():
            out[f + k] = v
        mp.mapping(6)  # accumulates certainties in the cells the batch touches
    out["travel_dist"] = np.asarray(travel, np.float32)
    return out



def gen_preprocess():
    """(9) SLAMDataset.preprocess_frame data path (dataset/slam_dataset.py:359-505): the two
    voxel_down_sample_torch passes, crop_frame, intrinsic_correct and deskewing, each called as
    the reference function on one synthetic scan.  roma is not installed here: deskewing runs
    with roma.rotmat_slerp replaced by exp(t*log(R)) through scipy (what roma computes), so the
    fixture pins the reference's surrounding arithmetic (timestamp normalisation, mid-pose
    shift, translation lerp), not roma's own rounding."""
    import importlib
    from scipy.spatial.transform import Rotation
    m = R.load()
    sys.path.insert(0, R.REF_ROOT)
    try:
        sd = importlib.import_module("dataset.slam_dataset")
    finally:
        sys.path.remove(R.REF_ROOT)
    tools = m["tools"]
    gen = torch.Generator().manual_seed(21)
    n = 30000
    pts = sheet_points(gen, n, 45.0, layers=4)
    pts[:, 2] += 1.5 * torch.randn(n, generator=gen)
    inten = torch.rand(n, 1, generator=gen)
    scan = torch.cat([pts, inten], 1).float()
    ts = torch.rand(n, generator=gen).float()
    out = dict(scan=t2n(scan), ts=t2n(ts), vox_down_m=np.float64(0.12), source_vox_down_m=np.float64(0.8),
               min_z=np.float64(-3.0), max_z=np.float64(2.5), min_range=np.float64(2.5), max_range=np.float64(40.0),
               correct_deg=np.float64(0.195))
    idx = tools.voxel_down_sample_torch(scan[:, :3], out["vox_down_m"].item())
    out["idx_train"] = t2n(idx)
    pc, pts_ts = scan[idx], ts[idx]
    pc, pts_ts = sd.crop_frame(pc, pts_ts, -3.0, 2.5, 2.5, 40.0)
    out["cropped"], out["cropped_ts"] = t2n(pc), t2n(pts_ts)
    pc = sd.intrinsic_correct(pc.clone(), 0.195)
    out["corrected"] = t2n(pc)
    idx2 = tools.voxel_down_sample_torch(pc[:, :3], out["source_vox_down_m"].item())
    out["idx_source"] = t2n(idx2)
    src, src_ts = pc[idx2][:, :3].clone(), pts_ts[idx2].clone()
    out["source"], out["source_ts"] = t2n(src), t2n(src_ts)
    pose = np.eye(4)
    pose[:3, :3] = Rotation.from_rotvec([0.01, -0.02, 0.06]).as_matrix()
    pose[:3, 3] = [1.1, 0.05, -0.02]
    out["last_odom_tran"] = pose

    def slerp(R0, R1, steps):
        rv = Rotation.from_matrix((R0.T @ R1).double().numpy()).as_rotvec()
        Rs = Rotation.from_rotvec(steps.double().numpy()[:, None] * rv[None, :]).as_matrix()
        return (R0.double() @ torch.from_numpy(Rs)).to(R1)

    tools.roma.rotmat_slerp = slerp
    out["deskewed"] = t2n(tools.deskewing(src.clone(), src_ts.clone(), torch.tensor(pose, dtype=torch.float32)))
    return out



def gen_mesher():
    """(10) Mesher.query_points (utils/mesher.py:40-164): bulk forward-only SDF queries + marching-cubes
    mask over a regular grid that reaches into empty space, global and local index space, both
    weighting modes.  Map geometry = the c2_wf / kitti_nwf fixtures (seeded build; checked)."""
    import importlib
    out = {}
    for case in ("c2_wf", "kitti_nwf"):
        m, cfg, dec, npts, gen = build(case)
        ref = dict(np.load(os.path.join(OUT, case + ".npz")))
        # geometry (positions, table, local map) is reproduced bit for bit; the reference's multi-threaded
        # pre-training is not, so the trained arrays travel with this fixture
        assert np.array_equal(ref["neural_points"], t2n(npts.neural_points)) and np.array_equal(ref["local_mask"], t2n(npts.local_mask))
        out[case + "_geo_features"] = t2n(npts.geo_features)
        out[case + "_local_geo_features"] = t2n(npts.local_geo_features.data)
        out[case + "_dec_flat"] = flat_decoder(dec)
        sys.path.insert(0, R.REF_ROOT)
        try:
            mm = importlib.import_module("utils.mesher")
        finally:
            sys.path.remove(R.REF_ROOT)
        mesher = mm.Mesher(cfg, npts, {"sdf": dec, "semantic": None, "color": None})
        ax = torch.arange(4.0, 26.0, 0.55)
        az = torch.arange(-3.2, 0.4, 0.45)
        grid = torch.stack(torch.meshgrid(ax, torch.arange(-9.0, 9.0, 0.55), az, indexing="ij"), -1).reshape(-1, 3).float()
        out[case + "_grid"] = t2n(grid)
        for loc in (False, True):
            sdf, _, _, mask = mesher.query_points(grid, 4096, True, False, False, True, query_locally=loc,
                                                  mask_min_nn_count=4, out_torch=True)
            out[f"{case}_sdf_{'local' if loc else 'global'}"] = t2n(sdf)
            out[f"{case}_mask_{'local' if loc else 'global'}"] = t2n(mask)
    return out



def gen_postloop():
    """(11) post-loop map maintenance: NeuralPoints.adjust_map, recreate_hash
    (kept_points, by timestamp and by certainty), prune_map (local / global) and
    Mapper.transform_data_pool, on a four-frame map with per-frame pose corrections."""
    from scipy.spatial.transform import Rotation
    m = R.load()
    cfg = R.make_config(voxel_size_m=0.4, buffer_size=40009, local_map_radius=20.0, local_map_travel_dist_ratio=1.0,
                        feature_std=0.05, track_on=True)
    npts = m["NeuralPoints"](cfg)
    gen = torch.Generator().manual_seed(31)
    travel = [0.0, 9.0, 18.0, 27.0]
    npts.travel_dist = torch.tensor(travel, dtype=torch.float32)
    for ts in range(4):
        pts = sheet_points(gen, 6000, 12.0, center=(9.0 * ts, 0.0))
        npts.update(pts, torch.tensor([9.0 * ts, 0.0, 0.0]), torch.eye(3), ts)
    P = npts.count()
    npts.point_certainties = torch.rand(P, generator=gen) * 4.0
    npts.point_ts_update = torch.randint(0, 4, (P,), generator=gen).int()
    out = dict(buffer_size=np.int64(cfg.buffer_size), resolution=np.float64(cfg.voxel_size_m), travel_dist=np.asarray(travel, np.float32),
               diff_travel_dist_local=np.float64(npts.diff_travel_dist_local), cur_ts=np.int64(3),
               neural_points=t2n(npts.neural_points), point_orientations=t2n(npts.point_orientations),
               point_ts_create=t2n(npts.point_ts_create), point_ts_update=t2n(npts.point_ts_update),
               point_certainties=t2n(npts.point_certainties), geo_features=t2n(npts.geo_features))
    rng = np.random.default_rng(4)
    pd = np.tile(np.eye(4), (4, 1, 1))
    for i in range(4):
        pd[i, :3, :3] = Rotation.from_rotvec(rng.normal(0, 0.03, 3)).as_matrix()
        pd[i, :3, 3] = rng.normal(0, 0.2, 3)
    out["pose_diff"] = pd
    pd_t = torch.tensor(pd, dtype=torch.float64)
    # Mapper.transform_data_pool (mapper.py:527-531)
    ds = R.FakeDataset(n_frames=4)
    dec = m["Decoder"](cfg, 32, 1, 1)
    mp = m["Mapper"](cfg, ds, npts, {"sdf": dec, "semantic": None, "color": None})
    n_pool = 20000
    mp.global_coord_pool = sheet_points(gen, n_pool, 30.0, center=(13.0, 0.0))
    mp.time_pool = torch.randint(0, 4, (n_pool,), generator=gen).int()
    out["pool_global"], out["pool_ts"] = t2n(mp.global_coord_pool), t2n(mp.time_pool)
    mp.transform_data_pool(pd_t)
    out["pool_global_after"] = t2n(mp.global_coord_pool)
    # prune masks (computed like prune_map, without mutating)
    npts.cur_ts = 3
    for name, gp in (("local", False), ("global", True)):
        import copy
        q = copy.deepcopy(npts)
        q.silence = True
        changed = q.prune_map(1.0, min_prune_count=50, global_prune=gp)
        out[f"prune_{name}_changed"] = np.bool_(changed)
        out[f"prune_{name}_points"] = t2n(q.neural_points)
        out[f"prune_{name}_geo"] = t2n(q.geo_features)
        out[f"prune_{name}_ts_create"] = t2n(q.point_ts_create)
    # adjust_map, then recreate_hash in both selection modes
    npts.adjust_map(pd_t)
    out["adj_points"], out["adj_orient"] = t2n(npts.neural_points), t2n(npts.point_orientations)
    for name, with_ts in (("ts", True), ("cert", False)):
        npts.recreate_hash(None, None, True, with_ts, 3)
        tab = npts.buffer_pt_index
        slots = torch.nonzero(tab >= 0).flatten()
        out[f"rehash_{name}_slots"], out[f"rehash_{name}_vals"] = t2n(slots), t2n(tab[slots])
    # use_mid_ts: the timestamp of a point is ((ts_create + ts_update) / 2).int() (adjust_map / recreate_hash)
    import copy
    cfg.use_mid_ts = True
    q = copy.deepcopy(npts)
    q.config = cfg
    q.neural_points, q.point_orientations = torch.from_numpy(out["neural_points"]), torch.from_numpy(out["point_orientations"])
    q.adjust_map(pd_t)
    out["mid_adj_points"], out["mid_adj_orient"] = t2n(q.neural_points), t2n(q.point_orientations)
    q.recreate_hash(None, None, True, True, 3)
    slots = torch.nonzero(q.buffer_pt_index >= 0).flatten()
    out["rehash_mid_slots"], out["rehash_mid_vals"] = t2n(slots), t2n(q.buffer_pt_index[slots])
    cfg.use_mid_ts = False
    # the final merge of a run (pin_slam.py:520-521): prune_map(..., 0, True) then recreate_hash(None, None, False, False)
    q = copy.deepcopy(npts)
    q.silence = True
    q.prune_map(1.0, 0, True)
    q.recreate_hash(None, None, False, False)
    slots = torch.nonzero(q.buffer_pt_index >= 0).flatten()
    out.update(merge_points=t2n(q.neural_points), merge_orient=t2n(q.point_orientations), merge_geo=t2n(q.geo_features),
               merge_ts_create=t2n(q.point_ts_create), merge_ts_update=t2n(q.point_ts_update),
               merge_cert=t2n(q.point_certainties), merge_slots=t2n(slots), merge_vals=t2n(q.buffer_pt_index[slots]))
    return out




def ref_state_from_fixture(m, name):
    """The reference's NeuralPoints + SDF Decoder rebuilt FROM a committed fixture (so that a new fixture can share the
    map of an old one instead of carrying 10 MB of arrays again): global arrays and table as recorded, then
    reset_local_map at the fixture's sensor position -- checked against the fixture's own local arrays."""
    d = dict(np.load(os.path.join(OUT, name + ".npz")))
    over, (levels, hidden) = CASES[name]
    cfg = R.make_config(local_map_radius=20.0, local_map_travel_dist_ratio=1.0, bs=512, feature_std=0.1,
                        gradient_decimation=10, track_on=True, **over)
    cfg.geo_mlp_level, cfg.geo_mlp_hidden_dim = levels, hidden
    npts = m["NeuralPoints"](cfg)
    npts.neural_points = torch.from_numpy(d["neural_points"].copy())
    npts.point_orientations = torch.from_numpy(d["point_orientations"].copy())
    npts.geo_features = torch.from_numpy(d["geo_features"].copy())
    npts.point_ts_create = torch.from_numpy(d["point_ts_create"].copy())
    npts.point_ts_update = torch.from_numpy(d["point_ts_update"].copy())
    npts.point_certainties = torch.from_numpy(d["point_certainties"].copy())
    npts.buffer_pt_index[torch.from_numpy(d["table_slots"])] = torch.from_numpy(d["table_vals"])
    npts.travel_dist = torch.from_numpy(d["travel_dist"].copy())
    npts.reset_local_map(torch.tensor([16.0, 0.0, 0.0]), torch.eye(3), int(d["cur_ts"]))
    assert np.array_equal(t2n(npts.local_neural_points), d["local_neural_points"])
    assert np.array_equal(t2n(npts.global2local), d["global2local"]) and np.array_equal(t2n(npts.local_mask), d["local_mask"])
    npts.local_geo_features.data.copy_(torch.from_numpy(d["local_geo_features"]))
    npts.local_point_certainties = torch.from_numpy(d["local_point_certainties"].copy())
    npts.local_point_ts_update = torch.from_numpy(d["local_point_ts_update"].copy())
    dec = m["Decoder"](cfg, hidden, levels, 1)
    sd = dec.state_dict()
    flat, o = torch.from_numpy(d["dec_flat"].copy()), 0
    for k_, v in sd.items():
        sd[k_] = flat[o:o + v.numel()].view_as(v).clone()
        o += v.numel()
    dec.load_state_dict(sd)
    return d, cfg, npts, dec


def gen_variants():
    """Small fixture ON THE MAPS of c2_wf / kitti_nwf (loaded back into the reference's classes): configuration branches the
    other fixtures leave out.
      * reg_dist_div_grad_norm (tracker.py:452-456): one registration_step per map with the residual divided by |grad|;
      * the SEMANTIC head (run_demo_sem.yaml; decoder.py:100-103, mapper.py:664-667 / 782-800, tracker.py:336-341): a
        semantic decoder of sem_class_count + 1 = 21 heads (seeded), Decoder.sem_label_prob on query features,
        Tracker.query_source_points(query_sem=True), and two Mapper.mapping iterations on fixed batches WITH labels
        (-1 / 0 / classes), sem_label_decimation 1 (weighted-first map) and 3 with freespace_label_on (per-neighbour map):
        per-iteration gradients of the geometry features, the SDF decoder and the semantic decoder, the NLL term, the total
        loss, the parameters after the two Adam steps."""
    m = R.load()
    out = {}
    for name in ("c2_wf", "kitti_nwf"):
        d, cfg, npts, dec = ref_state_from_fixture(m, name)
        tools = m["tools"]
        trk = m["Tracker"](cfg, npts, {"sdf": dec, "semantic": None, "color": None})
        # ---- registration_step with reg_dist_div_grad_norm
        cfg.reg_dist_div_grad_norm = True
        cur = torch.from_numpy(d["reg_cur"])
        step = trk.registration_step(cur.clone(), None, torch.zeros(len(cur)), None, cfg.reg_min_grad_norm, cfg.reg_max_grad_norm,
                                     cfg.reg_GM_dist_m, cfg.reg_GM_grad, cfg.reg_lm_lambda, False)
        out[f"{name}_ddgn_dT"] = t2n(step[0]); out[f"{name}_ddgn_valid_count"] = np.int64(step[4].shape[0])
        out[f"{name}_ddgn_residual_cm"] = np.float64(step[5])
        cfg.reg_dist_div_grad_norm = False
        # ---- semantic head
        cfg.semantic_on, cfg.sem_class_count, cfg.weight_s = True, 20, 1.0
        cfg.sem_label_decimation = 1 if cfg.weighted_first else 3
        cfg.freespace_label_on = not cfg.weighted_first
        S = cfg.sem_class_count + 1
        torch.manual_seed(77)
        sem = m["Decoder"](cfg, cfg.geo_mlp_hidden_dim, cfg.geo_mlp_level, S)
        with torch.no_grad():  # (default init on features of size 0.1 gives logits = the biases: scale the first layer and the head so that the argmax depends on the input)
            sem.layers[0].weight.mul_(12.0); sem.lout.weight.mul_(6.0)
        out[f"{name}_sem_dec_flat"] = flat_decoder(sem)
        out[f"{name}_sem_heads"] = np.int64(S)
        q = torch.from_numpy(d["query"])
        gf, _, w, nn, _ = npts.query_feature(q.clone(), training_mode=False, query_locally=True)
        out[f"{name}_sem_prob"] = t2n(sem.sem_label_prob(gf))   # [N, S] / [N, k, S]
        trk = m["Tracker"](cfg, npts, {"sdf": dec, "semantic": sem, "color": None})
        res = trk.query_source_points(q.clone(), cfg.infer_bs, True, False, False, False, query_sem=True, query_locally=True,
                                      mask_min_nn_count=cfg.track_mask_query_nn_k)
        out[f"{name}_sem_pred"] = t2n(res[4])
        # ---- two mapping iterations with the semantic term
        gen = torch.Generator().manual_seed(5 if cfg.weighted_first else 6)
        ds = R.FakeDataset(n_frames=3)
        mp = m["Mapper"](cfg, ds, npts, {"sdf": dec, "semantic": sem, "color": None})
        mp.determine_used_pose()
        bs, batches = cfg.bs, []
        for it in range(2):
            coord, label = surface_samples(gen, bs, 12.0, (16.0, 0.0), sigma=0.2)
            ts = torch.randint(0, 3, (bs,), generator=gen).int()
            w_ = torch.ones(bs)
            sl = torch.randint(-1, S, (bs,), generator=gen).int()   # -1 unlabelled, 0 free space, 1..20 classes
            batches.append((coord, label, ts, w_, sl))
            out[f"{name}_map_coord{it}"] = t2n(coord); out[f"{name}_map_label{it}"] = t2n(label)
            out[f"{name}_map_ts{it}"] = t2n(ts); out[f"{name}_map_sem{it}"] = t2n(sl)
        box = {"i": 0}

        def fake_get_batch(global_coord=False):
            c, l, t, w_, sl = batches[box["i"]]
            box["i"] += 1
            return c.clone(), l.clone(), t.clone(), None, sl.clone(), None, w_.clone()

        mp.get_batch = fake_get_batch
        grads, nll = [], []
        real_setup = m["mapper_mod"].setup_optimizer

        def spy_setup(*a, **k):
            opt = real_setup(*a, **k)
            real_step = opt.step

            def stepf(*aa, **kk):
                grads.append(dict(feat=t2n(npts.local_geo_features.grad),
                                  dec=np.concatenate([t2n(p.grad).ravel() for p in dec.parameters()]),
                                  sem=np.concatenate([t2n(p.grad).ravel() for p in sem.parameters()])))
                return real_step(*aa, **kk)

            opt.step = stepf
            return opt

        real_nll = torch.nn.NLLLoss

        class SpyNLL(real_nll):
            def forward(self, a, b):
                v = super().forward(a, b)
                nll.append(float(v.detach()))
                return v

        m["mapper_mod"].setup_optimizer = spy_setup
        torch.nn.NLLLoss = SpyNLL
        try:
            with LossSpy(m["mapper_mod"]) as spy:
                mp.mapping(2)
        finally:
            m["mapper_mod"].setup_optimizer = real_setup
            torch.nn.NLLLoss = real_nll
        out[f"{name}_map_loss_sdf"] = np.asarray(spy.sdf, np.float64); out[f"{name}_map_loss_total"] = np.asarray(spy.total, np.float64)
        out[f"{name}_map_loss_sem"] = np.asarray(nll, np.float64)
        for it, g in enumerate(grads):
            out[f"{name}_map_gfeat{it}"] = g["feat"]; out[f"{name}_map_gdec{it}"] = g["dec"]; out[f"{name}_map_gsem{it}"] = g["sem"]
        out[f"{name}_map_feat_after"] = t2n(npts.local_geo_features.data)
        out[f"{name}_map_dec_after"] = flat_decoder(dec)
        out[f"{name}_map_sem_after"] = flat_decoder(sem)
        for kname in ("sem_label_decimation", "freespace_label_on", "weight_s", "weight_e", "lr", "adam_eps", "gradient_decimation"):
            out[f"{name}_cfg_{kname}"] = np.float64(getattr(cfg, kname))
        out[f"{name}_map_eps"] = np.float64(cfg.voxel_size_m * cfg.num_grad_step_ratio)
    return out


def gen_process_sem():
    """Mapper.process_frame with per-point semantic labels (mapper.py:221, 280-283, 349-350; data_sampler.py:59-62,
    184-194): two frames of the `process` fixture's set-up; recorded: the scan, its labels, the random draws, and
    sem_label_pool after the sampler / the pool filter (the other pools are the `process` fixture's business)."""
    m = R.load()
    cfg = R.make_config(voxel_size_m=0.4, buffer_size=40009, local_map_radius=22.0, local_map_travel_dist_ratio=5.0,
                        bs=1024, bs_new_sample=256, feature_std=0.05, track_on=True, pool_capacity=20000,
                        pool_filter_freq=1, new_certainty_thre=1.0, surface_sample_range_m=0.25,
                        free_sample_end_dist_m=1.0, max_range=20.0, adaptive_iters=True, search_alpha=0.5, query_nn_k=6)
    cfg.window_radius = 13.0
    cfg.semantic_on, cfg.sem_class_count = True, 20
    torch.manual_seed(7)
    dec = m["Decoder"](cfg, 32, 1, 1)
    sem = m["Decoder"](cfg, 32, 1, 21)
    npts = m["NeuralPoints"](cfg)
    nfr = 2
    ds = R.FakeDataset(n_frames=nfr)
    mp = m["Mapper"](cfg, ds, npts, {"sdf": dec, "semantic": sem, "color": None})
    gen = torch.Generator().manual_seed(13)
    out = dict(n_frames=np.int64(nfr), buffer_size=np.int64(cfg.buffer_size), resolution=np.float64(cfg.voxel_size_m))
    for k in ("surface_sample_range_m", "surface_sample_n", "free_front_n", "free_behind_n", "free_sample_begin_ratio",
              "free_sample_end_dist_m", "dist_weight_on", "dist_weight_scale", "max_range", "behind_dropoff_on",
              "window_radius", "pool_capacity", "new_certainty_thre", "map_surface_ratio", "bs_new_sample"):
        out[k] = np.asarray(getattr(cfg, k))
    travel = [0.0]
    for ts in range(nfr):
        pose = np.eye(4)
        pose[:3, 3] = [6.0 * ts, 0.5 * ts, 0.1 * ts]
        if ts:
            travel.append(travel[-1] + float(np.linalg.norm(pose[:3, 3] - prev[:3, 3])))
        prev = pose
        ds.odom_poses[ts] = pose
        ds.processed_frame = ts
        npts.travel_dist = torch.tensor(travel + [0.0] * (nfr - len(travel)), dtype=torch.float32)
        scan = sheet_points(gen, 2000, 12.0)
        labels = torch.randint(0, 21, (scan.shape[0],), generator=gen).int()
        with _RngSpy() as spy:
            mp.process_frame(scan, labels, torch.tensor(pose, dtype=torch.float64), ts)
        f = f"f{ts}_"
        out[f + "scan"] = t2n(scan); out[f + "labels"] = t2n(labels); out[f + "pose"] = pose
        out[f + "rnd_surface"], out[f + "rnd_front"], out[f + "rnd_behind"] = (t2n(c[1]).reshape(-1) for c in spy.calls[:3])
        disc = [c[1] for c in spy.calls if c[0] == "randint"]
        out[f + "discard_index"] = t2n(disc[0]) if disc else np.zeros((0,), np.int64)
        out[f + "after_sem_label_pool"] = t2n(mp.sem_label_pool)
        out[f + "after_sdf_label_pool"] = t2n(mp.sdf_label_pool)
        out[f + "pool_sample_count"] = np.int64(mp.pool_sample_count); out[f + "cur_sample_count"] = np.int64(mp.cur_sample_count)
    out["travel_dist"] = np.asarray(travel, np.float32)
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    only = sys.argv[1] if len(sys.argv) > 1 else None
    for case in CASES:
        if only not in (None, case):
            continue
        d = gen_case(case)
        path = os.path.join(OUT, f"{case}.npz")
        np.savez_compressed(path, **d)
        print(case, "->", path, f"{os.path.getsize(path)/1e6:.2f} MB", "P =", d["neural_points"].shape[0],
              "M =", d["local_neural_points"].shape[0])
    if only in (None, "analytic_eik"):
        d = gen_analytic()
        path = os.path.join(OUT, "analytic_eik.npz")
        np.savez_compressed(path, **d)
        print("analytic_eik ->", path, f"{os.path.getsize(path)/1e6:.2f} MB", "losses", d["nwf_loss_total"], d["pgo_loss_total"], d["wf_loss_total"])
    if only in (None, "replica_color"):
        d = gen_color_case()
        path = os.path.join(OUT, "replica_color.npz")
        np.savez_compressed(path, **d)
        print("replica_color ->", path, f"{os.path.getsize(path)/1e6:.2f} MB")
    for name, col in (("process", False), ("process_color", True)):
        if only in (None, name):
            d = gen_process(color=col)
            path = os.path.join(OUT, f"{name}.npz")
            np.savez_compressed(path, **d)
            print(name, "->", path, f"{os.path.getsize(path)/1e6:.2f} MB", "pool", [int(d[f"f{t}_pool_sample_count"]) for t in range(int(d["n_frames"]))],
                  "new", [len(d[f"f{t}_new_idx"]) for t in range(int(d["n_frames"]))],
                  "discard", [len(d[f"f{t}_discard_index"]) for t in range(int(d["n_frames"]))],
                  "adaptive", [int(d[f"f{t}_adaptive_iter_offset"]) for t in range(int(d["n_frames"]))])
    if only in (None, "postloop"):
        d = gen_postloop()
        path = os.path.join(OUT, "postloop.npz")
        np.savez_compressed(path, **d)
        print("postloop ->", path, f"{os.path.getsize(path)/1e6:.2f} MB", "P", len(d["neural_points"]), "pruned",
              len(d["neural_points"]) - len(d["prune_local_points"]), len(d["neural_points"]) - len(d["prune_global_points"]))
    if only in (None, "mesher"):
        d = gen_mesher()
        path = os.path.join(OUT, "mesher.npz")
        np.savez_compressed(path, **d)
        print("mesher ->", path, f"{os.path.getsize(path)/1e6:.2f} MB", {k: (v.shape, float(np.mean(v != 0))) for k, v in d.items() if "mask" in k})
    if only in (None, "preprocess"):
        d = gen_preprocess()
        path = os.path.join(OUT, "preprocess.npz")
        np.savez_compressed(path, **d)
        print("preprocess ->", path, f"{os.path.getsize(path)/1e6:.2f} MB", "train", len(d["idx_train"]), "cropped",
              len(d["cropped"]), "source", len(d["idx_source"]))
    if only in (None, "variants"):
        d = gen_variants()
        path = os.path.join(OUT, "variants.npz")
        np.savez_compressed(path, **d)
        print("variants ->", path, f"{os.path.getsize(path)/1e6:.2f} MB", {k: v for k, v in d.items() if "loss" in k or "ddgn_valid" in k})
    if only in (None, "process_sem"):
        d = gen_process_sem()
        path = os.path.join(OUT, "process_sem.npz")
        np.savez_compressed(path, **d)
        print("process_sem ->", path, f"{os.path.getsize(path)/1e6:.2f} MB", [int(d[f"f{t}_pool_sample_count"]) for t in range(2)])
    if only not in (None, "update"):
        return
    d = gen_update()
    path = os.path.join(OUT, "update.npz")
    np.savez_compressed(path, **d)
    print("update ->", path, f"{os.path.getsize(path)/1e6:.2f} MB")


if __name__ == "__main__":
    sys.exit(main())
