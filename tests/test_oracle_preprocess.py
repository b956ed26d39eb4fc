"""Pin the oracle's restatement of the preprocess_frame data path (voxel down-sampling, crop_frame,
intrinsic_correct, deskewing) against fixtures recorded from the reference functions
(oracle/make_golden.py gen_preprocess).  CPU only."""
import numpy as np
import pytest

from oracle import pin_oracle as O
from tests import golden_util as G


@pytest.fixture(scope="module")
def pp():
    return G.load("preprocess")


def test_two_voxel_downsample_passes(pp):
    d = pp
    assert np.array_equal(O.voxel_down_sample(d["scan"][:, :3], d["vox_down_m"]), d["idx_train"])
    assert np.array_equal(O.voxel_down_sample(d["corrected"][:, :3], d["source_vox_down_m"]), d["idx_source"])


def test_crop_frame_bit_exact(pp):
    d = pp
    pc, ts = d["scan"][d["idx_train"]], d["ts"][d["idx_train"]]
    m = O.crop_frame_mask(pc, d["min_z"], d["max_z"], d["min_range"], d["max_range"])
    assert np.array_equal(pc[m], d["cropped"]) and np.array_equal(ts[m], d["cropped_ts"])
    assert 0.3 < m.mean() < 0.95


def test_intrinsic_correct(pp):
    d = pp
    got = O.intrinsic_correct(d["cropped"], d["correct_deg"])
    np.testing.assert_allclose(got, d["corrected"], rtol=1e-6, atol=1e-6)  # asin/sin/cos: libm vs sleef, last bits
    assert np.abs(got[:, :3] - d["cropped"][:, :3]).max() > 1e-3


def test_deskewing(pp):
    d = pp
    got = O.deskewing(d["source"], d["source_ts"], d["last_odom_tran"])
    np.testing.assert_allclose(got, d["deskewed"], rtol=0, atol=2e-5)
    assert np.abs(got - d["source"]).max() > 0.3  # half a metre of motion at the scan ends
