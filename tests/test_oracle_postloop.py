"""Pin the oracle's post-loop map maintenance (adjust_map, recreate_hash, prune_map,
transform_data_pool) against the reference fixture (oracle/make_golden.py gen_postloop).  CPU only."""
import numpy as np
import pytest

from oracle import pin_oracle as O
from tests import golden_util as G


@pytest.fixture(scope="module")
def pl():
    return G.load("postloop")


def test_transform_data_pool(pl):
    got = O.transform_batch(pl["pool_global"], pl["pose_diff"][pl["pool_ts"]])
    np.testing.assert_allclose(got, pl["pool_global_after"], rtol=0, atol=4e-6)
    assert np.abs(got - pl["pool_global"]).max() > 0.1


def test_adjust_map(pl):
    pos, q = O.adjust_map(pl["neural_points"], pl["point_orientations"], pl["point_ts_create"], pl["pose_diff"])
    np.testing.assert_allclose(pos, pl["adj_points"], rtol=0, atol=4e-6)
    np.testing.assert_allclose(q, pl["adj_orient"], rtol=0, atol=1e-6)
    assert np.abs(q[:, 1:]).max() > 0.005


@pytest.mark.parametrize("mode", ["ts", "cert"])
def test_recreate_hash(pl, mode):
    """Occupied slots and every slot with a single writer must equal the reference's table.  Where several
    voxels share a slot the reference's multi-threaded index_put_ keeps whichever thread wrote last
    (unspecified); ours keeps the last in sample order -- the reference's value must be one of the writers."""
    B = int(pl["buffer_size"])
    table, sel = O.recreate_hash(pl["adj_points"], pl["point_ts_create"], int(pl["cur_ts"]), pl["resolution"], B,
                                 certainties=pl["point_certainties"], with_ts=mode == "ts")
    slots = np.nonzero(table >= 0)[0]
    assert np.array_equal(slots, pl[f"rehash_{mode}_slots"])
    ref = np.full(B, -1, np.int64)
    ref[pl[f"rehash_{mode}_slots"]] = pl[f"rehash_{mode}_vals"]
    slot_of = O.hash_slots(O.grid_coords(pl["adj_points"][sel], pl["resolution"]), B)
    writers = np.bincount(slot_of, minlength=B)
    single = writers == 1
    assert np.array_equal(table[single], ref[single]) and single.sum() > 1000
    multi = np.nonzero(writers > 1)[0]
    assert len(multi) > 100
    for sl in multi[:300]:
        assert ref[sl] in sel[slot_of == sl] and table[sl] == sel[slot_of == sl][-1]


@pytest.mark.parametrize("mode", ["local", "global"])
def test_prune_map(pl, mode):
    m = O.prune_mask(pl["point_certainties"], pl["point_ts_update"], pl["travel_dist"], int(pl["cur_ts"]),
                     pl["diff_travel_dist_local"], 1.0, global_prune=mode == "global")
    assert bool(pl[f"prune_{mode}_changed"]) and m.sum() > 50
    assert np.array_equal(pl["neural_points"][~m], pl[f"prune_{mode}_points"])
    assert np.array_equal(pl["point_ts_create"][~m], pl[f"prune_{mode}_ts_create"])
    assert np.array_equal(pl["geo_features"][np.concatenate([~m, [True]])], pl[f"prune_{mode}_geo"])


def test_use_mid_ts(pl):
    """config.use_mid_ts (run_ncd*.yaml): adjust_map and recreate_hash by ((ts_create + ts_update) / 2).int()."""
    pos, q = O.adjust_map(pl["neural_points"], pl["point_orientations"], pl["point_ts_create"], pl["pose_diff"],
                          ts_update=pl["point_ts_update"])
    np.testing.assert_allclose(pos, pl["mid_adj_points"], rtol=0, atol=4e-6)
    np.testing.assert_allclose(q, pl["mid_adj_orient"], rtol=0, atol=1e-6)
    assert np.abs(pos - pl["adj_points"]).max() > 0.05
    B = int(pl["buffer_size"])
    table, sel = O.recreate_hash(pl["mid_adj_points"], pl["point_ts_create"], int(pl["cur_ts"]), pl["resolution"], B,
                                 ts_update=pl["point_ts_update"])
    assert np.array_equal(np.nonzero(table >= 0)[0], pl["rehash_mid_slots"])
    ref = np.full(B, -1, np.int64)
    ref[pl["rehash_mid_slots"]] = pl["rehash_mid_vals"]
    slot_of = O.hash_slots(O.grid_coords(pl["mid_adj_points"][sel], pl["resolution"]), B)
    single = np.bincount(slot_of, minlength=B) == 1
    assert np.array_equal(table[single], ref[single]) and single.sum() > 1000


def merged_reference(pl):
    return dict(positions=pl["merge_points"], orientations=pl["merge_orient"], ts_create=pl["merge_ts_create"],
                ts_update=pl["merge_ts_update"], certainties=pl["merge_cert"], geo_features=pl["merge_geo"])


def pruned_adjusted_map(pl):
    """The state the reference merged: the postloop map after adjust_map, then prune_map(1.0, 0, True)."""
    keep = ~O.prune_mask(pl["point_certainties"], pl["point_ts_update"], pl["travel_dist"], int(pl["cur_ts"]),
                         pl["diff_travel_dist_local"], 1.0, global_prune=True)
    return dict(positions=pl["adj_points"][keep], orientations=pl["adj_orient"][keep], ts_create=pl["point_ts_create"][keep],
                ts_update=pl["point_ts_update"][keep], certainties=pl["point_certainties"][keep],
                geo_features=pl["geo_features"][np.concatenate([keep, [True]])])


def test_final_merge(pl):
    """The end of a run (pin_slam.py:520-521): recreate_hash(None, None, False, False) keeps one point per voxel, the
    most certain one, and re-indexes the table."""
    B = int(pl["buffer_size"])
    got, table = O.merge_map(pruned_adjusted_map(pl), 0, pl["resolution"], B, with_ts=False)
    ref = merged_reference(pl)
    for k in ref:
        assert np.array_equal(got[k], ref[k]), k
    assert len(ref["positions"]) < 0.9 * len(pl["neural_points"])
    assert np.array_equal(np.nonzero(table >= 0)[0], pl["merge_slots"])
    rt = np.full(B, -1, np.int64)
    rt[pl["merge_slots"]] = pl["merge_vals"]
    slot_of = O.hash_slots(O.grid_coords(ref["positions"], pl["resolution"]), B)
    single = np.bincount(slot_of, minlength=B) == 1
    assert np.array_equal(table[single], rt[single]) and single.sum() > 1000
