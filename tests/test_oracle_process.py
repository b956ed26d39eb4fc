"""Pin the oracle's restatement of the Mapper.process_frame data path (DataSampler.sample, pool
window / discard, query_certainty, new-sample index) against fixtures recorded from the real
reference (oracle/make_golden.py gen_process).  CPU only."""
import numpy as np
import pytest

from oracle import pin_oracle as O
from tests import golden_util as G

POOLS = ("coord_pool", "global_coord_pool", "sdf_label_pool", "weight_pool", "time_pool")


@pytest.fixture(scope="module", params=["process", "process_color"])
def pg(request):
    return G.load(request.param)


def sampler_kw(d):
    return dict(surface_range=d["surface_sample_range_m"], surface_n=int(d["surface_sample_n"]),
                front_n=int(d["free_front_n"]), behind_n=int(d["free_behind_n"]),
                free_begin_ratio=d["free_sample_begin_ratio"], free_end_dist=d["free_sample_end_dist_m"],
                dist_weight_on=bool(d["dist_weight_on"]), dist_weight_scale=d["dist_weight_scale"],
                max_range=d["max_range"], behind_dropoff_on=bool(d["behind_dropoff_on"]))


def frame_pool_before(d, t, names):
    """Pools as Mapper.process_frame sees them right before its filter step: the previous
    frame's pools + this frame's samples (mapper.py:275-300)."""
    f = f"f{t}_"
    new = {"coord_pool": d[f + "s_coord"], "global_coord_pool": O.transform_points(d[f + "s_coord"], d[f + "pose"]),
           "sdf_label_pool": d[f + "s_label"], "weight_pool": d[f + "s_weight"],
           "time_pool": np.full(len(d[f + "s_label"]), t, np.int32)}
    if "color_pool" in names:
        new["color_pool"] = d[f + "s_color"]
    if t == 0:
        return new
    return {k: np.concatenate([d[f"f{t-1}_after_{k}"], new[k]], 0) for k in names}


def test_sample_rays_bit_exact(pg):
    d = pg
    for t in range(int(d["n_frames"])):
        f = f"f{t}_"
        scan = d[f + "scan"]
        col = scan[:, 3:] if scan.shape[1] > 3 else None
        coord, label, color, weight = O.sample_rays(scan[:, :3], col, d[f + "rnd_surface"], d[f + "rnd_front"],
                                                    d[f + "rnd_behind"], **sampler_kw(d))
        assert np.array_equal(coord.view(np.uint32), d[f + "s_coord"].view(np.uint32))
        assert np.array_equal(label.view(np.uint32), d[f + "s_label"].view(np.uint32))
        assert np.array_equal(weight.view(np.uint32), d[f + "s_weight"].view(np.uint32))
        if col is not None:
            assert np.array_equal(color, d[f + "s_color"])
        assert (weight < 0).sum() == len(scan) * (int(d["free_front_n"]) + int(d["free_behind_n"]))


def test_pool_window_and_discard(pg):
    d = pg
    names = POOLS + (("color_pool",) if "f0_s_color" in d else ())
    discarded = 0
    for t in range(int(d["n_frames"])):
        f = f"f{t}_"
        before = frame_pool_before(d, t, names)
        mask = O.pool_filter_mask(before["global_coord_pool"], d[f + "pose"][:3, 3], d["window_radius"],
                                  int(d["pool_capacity"]), d[f + "discard_index"])
        discarded += len(d[f + "discard_index"])
        assert int(mask.sum()) == d[f + "pool_sample_count"]
        n_cur = len(d[f + "s_label"])
        assert int(mask[-n_cur:].sum()) == d[f + "cur_sample_count"]
        for k in names:
            got, ref = before[k][mask], d[f + "after_" + k]
            if k == "global_coord_pool":  # sgemm of the reference vs the oracle's matmul: last-bit differences
                np.testing.assert_allclose(got, ref, rtol=0, atol=4e-6)
            else:
                assert np.array_equal(got, ref), k
    assert discarded > 0  # the capacity branch is exercised


def test_query_certainty_and_new_index(pg):
    d = pg
    seen = 0
    for t in range(int(d["n_frames"])):
        f = f"f{t}_"
        table = np.full(int(d["buffer_size"]), -1, np.int64)
        table[d[f + "qc_table_slots"]] = d[f + "qc_table_vals"]
        cur = int(d[f + "cur_sample_count"])
        q = d[f + "after_global_coord_pool"][-cur:]
        cert = O.query_certainty(q, table, d[f + "qc_positions"], d[f + "qc_certainties"], d["resolution"])
        assert np.array_equal(cert, d[f + "qc_out"])
        seen += int((cert > 0).sum())
        idx = O.new_sample_index(cert, d[f + "after_sdf_label_pool"][-cur:], d["new_certainty_thre"],
                                 d["surface_sample_range_m"], offset=int(d[f + "pool_sample_count"]) - cur)
        assert np.array_equal(idx, d[f + "new_idx"])
        off = O.adaptive_iter_offset(len(idx), cur, t, adaptive_iters=True, ratio_less=d["new_sample_ratio_less"],
                                     ratio_more=d["new_sample_ratio_more"], ratio_restart=d["new_sample_ratio_restart"],
                                     freeze_after_frame=int(d["freeze_after_frame"]))
        assert off == d[f + "adaptive_iter_offset"]
    assert seen > 1000  # certainties are not all zero (mapping ran between the frames)
