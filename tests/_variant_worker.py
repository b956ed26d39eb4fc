"""Worker of tests/test_gpu_variants.py: one kNN + GN pass on a small synthetic map under the environment it is
started with (kernel variants are chosen by environment variables read once per process); writes an .npz."""
import os, sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pin_slam_amd import ops, synth  # noqa: E402
from pin_slam_amd._lib import GnParams  # noqa: E402


def main(out, hidden, levels, orient):
    m = synth.build_map(layers=2, radius=20.0, raw_per_layer=120_000)
    P = len(m.positions)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    pos = dev(m.positions)
    pos4 = torch.empty((P, 4), dtype=torch.float32, device="cuda")
    ops.pack_positions(pos, torch.zeros(P, dtype=torch.int32, device="cuda"), pos4)
    dx, mv = ops.search_neighborhood(2, 0.5, 0.4)
    g2l = torch.arange(P + 1, dtype=torch.int32, device="cuda"); g2l[-1] = -1
    st = ops.SearchState(table=dev(m.table), pos4=pos4, cand_off=dev(ops.candidate_offsets(dx, m.buffer_size)), n_points=P,
                         resolution=0.4, max_valid_dist2=mv, travel_dist=torch.zeros(1, device="cuda"), cur_ts=0,
                         diff_travel_dist_local=410.0, global2local=g2l)
    rng = np.random.default_rng(5)
    quat = None
    if orient:
        q = rng.standard_normal((P + 1, 4)).astype(np.float32)
        quat = dev(q / np.linalg.norm(q, axis=1, keepdims=True))
    fs = ops.FieldState(feats=dev(m.features), dec=dev(synth.init_decoder(hidden, levels)), k=8, hidden=hidden, levels=levels,
                        weighted_first=True, sdf_scale=0.055, certainty=torch.zeros(P, device="cuda"), pos=pos, orient=quat)
    scan = dev(synth.make_scan(m, n=20_011, radius=18.0))
    gp = GnParams(); gp.valid_nn_k = 6; gp.min_grad_norm = 1e-5; gp.max_grad_norm = 1e3; gp.max_sdf_std = 0.25  # random-init decoder: tiny gradients
    gp.gm_dist = 0.3; gp.gm_grad = 0.1
    T = np.eye(4); T[:3, 3] = (0.03, -0.02, 0.01)
    bricks = ops.BrickCache(dx, 2).build(st, wait=True)
    nbr, nn, cur = ops.knn_query(st, scan, 8, pose=T, bricks=bricks)
    sums, sdf, grad = ops.gn_accumulate(fs, gp, cur, nbr, nn, want_points=True)
    torch.cuda.synchronize()
    np.savez(out, nbr=nbr.cpu().numpy().view(np.int32), nn=nn.cpu().numpy(), cur=cur.cpu().numpy(),
             sums=sums.cpu().numpy().sum(0), sdf=sdf.cpu().numpy(), grad=grad.cpu().numpy())


def query(out):
    """pin_sdf_query / pin_color_query over several field shapes on a small synthetic map, under the environment the process
    was started with (PIN_QUERY_QUAD picks the tile kernels or the thread-per-query kernels, read once per process)."""
    m = synth.build_map(layers=2, radius=20.0, raw_per_layer=120_000)
    P = len(m.positions)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    pos = dev(m.positions)
    pos4 = torch.empty((P, 4), dtype=torch.float32, device="cuda")
    ops.pack_positions(pos, torch.zeros(P, dtype=torch.int32, device="cuda"), pos4)
    dx, mv = ops.search_neighborhood(2, 0.5, 0.4)
    g2l = torch.arange(P + 1, dtype=torch.int32, device="cuda"); g2l[-1] = -1
    st = ops.SearchState(table=dev(m.table), pos4=pos4, cand_off=dev(ops.candidate_offsets(dx, m.buffer_size)), n_points=P,
                         resolution=0.4, max_valid_dist2=mv, travel_dist=torch.zeros(1, device="cuda"), cur_ts=0,
                         diff_travel_dist_local=410.0, global2local=g2l)
    rng = np.random.default_rng(11)
    q4 = rng.standard_normal((P + 1, 4)).astype(np.float32)
    quat = dev(q4 / np.linalg.norm(q4, axis=1, keepdims=True))
    cert = dev(rng.uniform(0.0, 5.0, P).astype(np.float32))
    # queries: a scan on the surface, points off the surface (fewer neighbours) and a few with none at all
    scan = synth.make_scan(m, n=6_001, radius=18.0)
    off = scan[:1500] + rng.normal(0, 0.35, (1500, 3)).astype(np.float32)
    off[:, 2] += (rng.uniform(0.9, 2.3, 1500) * rng.choice([-1.0, 1.0], 1500)).astype(np.float32)  # out to the edge of the search radius
    far = scan[:37] + np.float32(500.0)
    q = dev(np.concatenate([scan, off, far]).astype(np.float32))
    nbr, nn, _ = ops.knn_query(st, q, 8)
    res = {"nn": nn.cpu().numpy()}
    def big(h, l, od=1):  # a decoder with outputs and gradients well above rounding noise
        w = synth.init_decoder(h, l, out_dim=od)
        return dev((w + 0.2 * rng.standard_normal(w.shape)).astype(np.float32))
    cases = [("wf_64x4", 64, 4, True, None), ("wf_32x2_pgo", 32, 2, True, quat), ("wf_64x1", 64, 1, True, None),
             ("nwf_64x1", 64, 1, False, None), ("nwf_32x1_pgo", 32, 1, False, quat)]
    for tag, h, l, wf, ori in cases:
        fs = ops.FieldState(feats=dev(m.features), dec=big(h, l), k=8, hidden=h, levels=l, weighted_first=wf, sdf_scale=0.055,
                            certainty=cert, pos=pos, orient=ori)
        for staged in (False, True):
            if staged:
                fs.stage_decoder()
            sdf, grad, std, ce = ops.sdf_query(fs, q, nbr, nn)
            sdf0, _, _, _ = ops.sdf_query(fs, q, nbr, nn, grad=False, std=False, certainty=False)
            k = tag + ("_staged" if staged else "")
            res.update({k + "_sdf": sdf.cpu().numpy(), k + "_grad": grad.cpu().numpy(), k + "_std": std.cpu().numpy(),
                        k + "_cert": ce.cpu().numpy(), k + "_sdf_fwd": sdf0.cpu().numpy()})
    for tag, h, l, ori in (("col_64x2", 64, 2, None), ("col_32x1_pgo", 32, 1, quat)):
        fc = ops.FieldState(feats=dev(m.features), dec=big(h, l, 3), k=8, hidden=h,
                            levels=l, weighted_first=True, sdf_scale=1.0, certainty=None, pos=pos, orient=ori, out_dim=3)
        col, val, g = ops.color_query(fc, q, nbr, nn)
        col0, val0, _ = ops.color_query(fc, q, nbr, nn, want_grad=False)
        res.update({tag + "_col": col.cpu().numpy(), tag + "_val": val.cpu().numpy(), tag + "_grad": g.cpu().numpy(),
                    tag + "_col_fwd": col0.cpu().numpy(), tag + "_val_fwd": val0.cpu().numpy()})
    torch.cuda.synchronize()
    np.savez(out, **res)


def fixture(out, case):
    """Per-point outputs and sums of the GN tile kernel on a golden fixture's query set (local index space)."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from tests import golden_util as G
    from tests import gpu_util as U
    d = G.load(case)
    st, fs = U.search_state(d), U.field_state(d)
    q = U.dev(d["query"])
    nbr, nn, _ = ops.knn_query(st, q, int(d["query_nn_k"]))
    gp = GnParams()
    gp.valid_nn_k = int(d["track_mask_query_nn_k"])
    gp.min_grad_norm, gp.max_grad_norm = d["cfg_reg_min_grad_norm"], d["cfg_reg_max_grad_norm"]
    gp.max_sdf_std = d["cfg_surface_sample_range_m"] * d["cfg_max_sdf_std_ratio"]
    gp.gm_dist, gp.gm_grad = d["cfg_reg_GM_dist_m"], d["cfg_reg_GM_grad"]
    sums, sdf, grad = ops.gn_accumulate(fs, gp, q, nbr, nn, want_points=True)
    torch.cuda.synchronize()
    np.savez(out, nn=nn.cpu().numpy(), sums=sums.cpu().numpy().sum(0), sdf=sdf.cpu().numpy(), grad=grad.cpu().numpy())


def color(out):
    """Sums of the GN kernel with the colour term (photometric rows / consistency weight) on the replica_color fixture."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from tests import golden_util as G
    from tests import gpu_util as U
    d = G.load("replica_color")
    import dataclasses
    st, fs = U.search_state(d), U.field_state(d)
    fc = dataclasses.replace(fs, feats=U.dev(d["local_color_features"]), dec=U.dev(d["cdec_flat"]),
                             hidden=int(d["cdec_hidden"]), levels=int(d["cdec_levels"]), out_dim=3)
    src = U.dev(d["reg_src"])
    nbr, nn, cur = ops.knn_query(st, src, int(d["query_nn_k"]), pose=d["reg_Tinit"])
    gp = GnParams()
    gp.valid_nn_k = int(d["track_mask_query_nn_k"])
    gp.min_grad_norm, gp.max_grad_norm = d["cfg_reg_min_grad_norm"], d["cfg_reg_max_grad_norm"]
    gp.max_sdf_std = d["surface_sample_range_m"] * d["cfg_max_sdf_std_ratio"]
    gp.gm_dist, gp.gm_grad = d["cfg_reg_GM_dist_m"], d["cfg_reg_GM_grad"]
    res = {}
    for tag in ("photo", "consist"):
        ct, keep = ops.color_term(fc, U.dev(d["reg_colors"]), photometric=(tag == "photo"), photo_weight=d["photometric_loss_weight"])
        sums, _, _ = ops.gn_accumulate(fs, gp, cur, nbr, nn, color=ct)
        res[tag] = sums.cpu().numpy().sum(0)
    torch.cuda.synchronize()
    np.savez(out, **res)


if __name__ == "__main__":
    if sys.argv[2] == "color":
        color(sys.argv[1])
    elif sys.argv[2] == "query":
        query(sys.argv[1])
    elif sys.argv[2] == "fixture":
        fixture(sys.argv[1], sys.argv[3])
    else:
        main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
