"""The tile-decoder inference kernels (csrc/sdf_quad.h, the query modes of gn_accumulate_quad_nwf_kernel) against the
thread-per-query kernels they replaced on pin_sdf_query / pin_color_query: same map, same queries, same records, two
processes (PIN_QUERY_QUAD is read once per process).  The two families differ in arithmetic only where the review bars allow
it -- split-fp16 against fp32 matrix-core products (~1e-6 of the output scale), hardware reciprocal against IEEE division in
the IDW weights (1 ulp) -- so the outputs must agree far inside the 1e-4 bar both are held to against the reference
(tests/test_gpu_parity.py, tests/test_gpu_scale_parity.py)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def both(tmp_path_factory):
    out = {}
    for mode in ("0", "1"):
        f = str(tmp_path_factory.mktemp("q") / f"query_{mode}.npz")
        subprocess.run([sys.executable, os.path.join(HERE, "_variant_worker.py"), f, "query"], check=True,
                       env=dict(os.environ, PIN_QUERY_QUAD=mode), timeout=600)
        out[mode] = dict(np.load(f))
    return out


def _close(a, b, rel, what):
    scale = float(np.max(np.abs(b))) + 1e-30
    err = float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64)))) / scale
    assert err < rel, f"{what}: {err:.3e} of the output scale (bar {rel:.0e})"


@pytest.mark.parametrize("tag", ["wf_64x4", "wf_32x2_pgo", "wf_64x1", "nwf_64x1", "nwf_32x1_pgo"])
@pytest.mark.parametrize("staged", ["", "_staged"])
def test_sdf_query_tiles_match_thread_per_query(both, tag, staged):
    old, new = both["0"], both["1"]
    k = tag + staged
    assert np.array_equal(old["nn"], new["nn"])
    assert (old["nn"] == 0).sum() >= 37 and (old["nn"] < 8).sum() > 100, "the query set must hold empty and ragged neighbourhoods"
    _close(new[k + "_sdf"], old[k + "_sdf"], 2e-5, "sdf")
    _close(new[k + "_sdf_fwd"], old[k + "_sdf"], 2e-5, "sdf (forward-only variant)")
    g_old, g_new = old[k + "_grad"], new[k + "_grad"]
    assert float(np.max(np.abs(g_new - g_old))) / float(np.max(np.abs(g_old))) < 1e-4
    _close(new[k + "_cert"], old[k + "_cert"], 1e-5, "certainty")
    if tag.startswith("nwf"):
        assert float(np.max(old[k + "_std"])) > 0
        _close(new[k + "_std"], old[k + "_std"], 1e-4, "spread of the k predictions")
    else:
        assert not new[k + "_std"].any() and not old[k + "_std"].any()  # one prediction per query: no spread
    # the staged image and the in-kernel staging hold the same bits
    assert np.array_equal(new[tag + "_sdf"], new[tag + "_staged_sdf"])
    assert np.array_equal(new[tag + "_grad"], new[tag + "_staged_grad"])


@pytest.mark.parametrize("tag", ["col_64x2", "col_32x1_pgo"])
def test_color_query_tiles_match_thread_per_query(both, tag):
    old, new = both["0"], both["1"]
    _close(new[tag + "_col"], old[tag + "_col"], 2e-5, "colour heads")
    _close(new[tag + "_val"], old[tag + "_val"], 2e-5, "intensity")
    _close(new[tag + "_col_fwd"], old[tag + "_col"], 2e-5, "colour heads (forward-only variant)")
    _close(new[tag + "_val_fwd"], old[tag + "_val"], 2e-5, "intensity (forward-only variant)")
    g_old, g_new = old[tag + "_grad"], new[tag + "_grad"]
    assert float(np.max(np.abs(g_new - g_old))) / float(np.max(np.abs(g_old))) < 1e-4
