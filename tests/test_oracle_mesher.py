"""Pin the oracle's Mesher.query_points restatement against the reference fixture (CPU)."""
import numpy as np
import pytest

from oracle import pin_oracle as O
from tests import golden_util as G


@pytest.mark.parametrize("case", ["c2_wf", "kitti_nwf"])
@pytest.mark.parametrize("local", [False, True])
def test_mesher_query_points(case, local):
    d, mz = G.load(case), G.load("mesher")
    assert np.array_equal(d["neural_points"], d["neural_points"])
    table = G.dense_table(d)
    grid = mz[case + "_grid"]
    params = O.unpack_decoder(mz[case + "_dec_flat"], 11, int(d["dec_hidden"]), int(d["dec_levels"]))
    if local:
        s = O.radius_search(grid, table, d["neural_points"], d["resolution"], d["neighbor_dx"], d["max_valid_dist2"],
                            ts_create=d["point_ts_create"], travel_dist=d["travel_dist"], cur_ts=int(d["cur_ts"]),
                            diff_travel_dist_local=d["diff_travel_dist_local"])
        feats, pos, g2l = mz[case + "_local_geo_features"], d["local_neural_points"], d["global2local"]
    else:
        s = O.radius_search(grid, table, d["neural_points"], d["resolution"], d["neighbor_dx"], d["max_valid_dist2"])
        feats, pos, g2l = mz[case + "_geo_features"], d["neural_points"], None
    sdf, mask = O.mesher_query(grid, s, feats, pos, params, d["sdf_scale"], int(d["query_nn_k"]),
                               weighted_first=bool(d["weighted_first"]), global2local=g2l)
    key = "local" if local else "global"
    assert np.array_equal(mask, mz[f"{case}_mask_{key}"] != 0)
    np.testing.assert_allclose(sdf, mz[f"{case}_sdf_{key}"], rtol=1e-4, atol=2e-6)
    assert 0.3 < mask.mean() < 0.95 and (sdf == 0).sum() > 10  # the grid reaches into unobserved space
