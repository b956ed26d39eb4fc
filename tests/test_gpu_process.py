"""GPU parity of the Mapper.process_frame data path kernels (K12-K14) through the C ABI against
the fixtures recorded from the reference (tests/golden/process*.npz) and the oracle."""
import numpy as np
import pytest
import torch

from oracle import pin_oracle as O
from tests import golden_util as G
from tests.test_oracle_process import POOLS, frame_pool_before

pytestmark = pytest.mark.gpu


class _Cfg:
    pass


def _cfg(d):
    c = _Cfg()
    for k in ("surface_sample_range_m", "surface_sample_n", "free_front_n", "free_behind_n", "free_sample_begin_ratio",
              "free_sample_end_dist_m", "dist_weight_on", "dist_weight_scale", "max_range", "behind_dropoff_on"):
        setattr(c, k, d[k])
    return c


@pytest.fixture(scope="module", params=["process", "process_color"])
def pg(request):
    return G.load(request.param)


def _names(d):
    return POOLS + (("color_pool",) if "f0_s_color" in d else ())


FIELD_OF = dict(coord_pool="coord", global_coord_pool="global_coord", sdf_label_pool="sdf_label", weight_pool="weight",
                time_pool="ts", color_pool="color")


def test_sampler_and_pool_filter_follow_reference(pg):
    """Four frames of append -> window/discard/compaction; every pool array after every frame
    equals the reference's (bit-exact except the transformed coordinates: sgemm vs fma order)."""
    from pin_slam_amd import pool as P
    d = pg
    C = 3 if "f0_s_color" in d else 0
    pool = P.SamplePool(color_channels=C, capacity=1024)
    for t in range(int(d["n_frames"])):
        f = f"f{t}_"
        scan = torch.from_numpy(d[f + "scan"]).cuda()
        rnd = tuple(torch.from_numpy(d[f + k]).cuda() for k in ("rnd_surface", "rnd_front", "rnd_behind"))
        n0 = pool.n
        n_new = pool.append_samples(scan, P.sample_params(_cfg(d), d[f + "pose"], t), rnd=rnd)
        assert n_new == len(d[f + "s_label"])
        tail = {k: pool.view(FIELD_OF[k])[n0:].cpu().numpy() for k in _names(d)}
        assert np.array_equal(tail["coord_pool"].view(np.uint32), d[f + "s_coord"].view(np.uint32))
        assert np.array_equal(tail["sdf_label_pool"].view(np.uint32), d[f + "s_label"].view(np.uint32))
        assert np.array_equal(tail["weight_pool"].view(np.uint32), d[f + "s_weight"].view(np.uint32))
        assert (tail["time_pool"] == t).all()
        if C:
            assert np.array_equal(tail["color_pool"], d[f + "s_color"])
        np.testing.assert_allclose(tail["global_coord_pool"], O.transform_points(d[f + "s_coord"], d[f + "pose"]),
                                   rtol=0, atol=4e-6)
        disc = torch.from_numpy(d[f + "discard_index"]).cuda()
        n_pool, n_cur = pool.filter(d[f + "pose"][:3, 3], d["window_radius"], int(d["pool_capacity"]),
                                    discard_index=disc if disc.numel() else None)
        assert (n_pool, n_cur) == (d[f + "pool_sample_count"], d[f + "cur_sample_count"])
        for k in _names(d):
            got, ref = pool.view(FIELD_OF[k]).cpu().numpy(), d[f + "after_" + k]
            if k == "global_coord_pool":
                np.testing.assert_allclose(got, ref, rtol=0, atol=4e-6)
            else:
                assert np.array_equal(got, ref), (k, t)


def test_query_certainty_and_new_index(pg):
    from pin_slam_amd import ops
    d = pg
    B = int(d["buffer_size"])
    dx, mv = ops.search_neighborhood(1, 0.0, d["resolution"])
    cand = torch.from_numpy(ops.candidate_offsets(dx, B)).cuda()
    for t in range(int(d["n_frames"])):
        f = f"f{t}_"
        table = np.full(B, -1, np.int32)
        table[d[f + "qc_table_slots"]] = d[f + "qc_table_vals"]
        pos = torch.from_numpy(d[f + "qc_positions"]).cuda()
        P_ = pos.shape[0]
        pos4 = torch.zeros((P_, 4), dtype=torch.float32, device="cuda")
        ops.pack_positions(pos, torch.zeros(P_, dtype=torch.int32, device="cuda"), pos4, 0, P_)
        st = ops.SearchState(table=torch.from_numpy(table).cuda(), pos4=pos4, cand_off=cand, n_points=P_,
                             resolution=d["resolution"], max_valid_dist2=mv)
        cur = int(d[f + "cur_sample_count"])
        q = torch.from_numpy(d[f + "after_global_coord_pool"][-cur:].copy()).cuda()
        cert = ops.query_certainty(st, torch.from_numpy(d[f + "qc_certainties"]).cuda(), q)
        assert np.array_equal(cert.cpu().numpy(), d[f + "qc_out"])
        lab = torch.from_numpy(d[f + "after_sdf_label_pool"][-cur:].copy()).cuda()
        idx, cnt = ops.new_sample_index(cert, lab, d["new_certainty_thre"], d["surface_sample_range_m"] * 3.0,
                                        offset=int(d[f + "pool_sample_count"]) - cur)
        c = int(cnt.item())
        assert np.array_equal(idx[:c].cpu().numpy(), d[f + "new_idx"])


def test_sampler_large_random_matches_oracle():
    """100k-point scan with torch-drawn noise (the product path's RNG use): bit-exact vs the oracle."""
    from pin_slam_amd import pool as P
    from pin_slam_amd.config import PinConfig
    torch.manual_seed(3)
    cfg = PinConfig()
    N = 100_000
    scan = (torch.rand(N, 3, device="cuda") - 0.5) * torch.tensor([80.0, 80.0, 6.0], device="cuda")
    pose = np.eye(4); pose[:3, 3] = [1.0, -2.0, 0.5]
    g = torch.Generator(device="cuda").manual_seed(5)
    rnd = (torch.randn(N * 3, 1, device="cuda", generator=g), torch.rand(N * 2, 1, device="cuda", generator=g),
           torch.rand(N * 1, 1, device="cuda", generator=g))
    pool = P.SamplePool(capacity=1024)
    pool.append_samples(scan, P.sample_params(cfg, pose, 4), rnd=rnd)
    coord, label, _, weight = O.sample_rays(scan.cpu().numpy(), None, *(r.cpu().numpy().reshape(-1) for r in rnd),
                                            surface_range=cfg.surface_sample_range_m, surface_n=3, front_n=2, behind_n=1,
                                            free_begin_ratio=cfg.free_sample_begin_ratio, free_end_dist=cfg.free_sample_end_dist_m,
                                            dist_weight_on=cfg.dist_weight_on, dist_weight_scale=cfg.dist_weight_scale,
                                            max_range=cfg.max_range)
    assert np.array_equal(pool.view("coord").cpu().numpy().view(np.uint32), coord.view(np.uint32))
    assert np.array_equal(pool.view("sdf_label").cpu().numpy().view(np.uint32), label.view(np.uint32))
    assert np.array_equal(pool.view("weight").cpu().numpy().view(np.uint32), weight.view(np.uint32))
