"""Kernel variants against each other (the variant is an environment variable read once per process, so every
case runs tests/_variant_worker.py in a subprocess):
  * PIN_MLP=f32 (fp32 MFMA) vs the default split-fp16 decoder: SDF / gradient within 2e-6 absolute (both are
    ~1e-7 from a double reference, scripts/decoder_bench.hip), Gauss-Newton sums within 1e-5 relative."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(tmp_path, name, env, hidden=64, levels=4, orient=0):
    out = str(tmp_path / (name + ".npz"))
    e = dict(os.environ)
    e.update(env)
    subprocess.run([sys.executable, os.path.join(HERE, "_variant_worker.py"), out, str(hidden), str(levels), str(orient)],
                   check=True, env=e, timeout=600)
    return np.load(out)


@pytest.mark.parametrize("hidden,levels,orient", [(64, 4, 0), (64, 4, 1), (32, 2, 0), (64, 2, 1), (64, 1, 0), (32, 3, 1)])
def test_split_fp16_decoder_matches_fp32_mfma(tmp_path, hidden, levels, orient):
    a = _run(tmp_path, "f32", {"PIN_MLP": "f32"}, hidden, levels, orient)
    b = _run(tmp_path, "h2", {"PIN_MLP": "h2"}, hidden, levels, orient)
    assert np.array_equal(a["nbr"], b["nbr"])
    assert np.abs(a["sdf"] - b["sdf"]).max() < 2e-6 * max(1.0, np.abs(a["sdf"]).max())
    assert np.abs(a["grad"] - b["grad"]).max() < 2e-6 * max(1.0, np.abs(a["grad"]).max())
    n_valid = a["sums"][29]
    assert n_valid > 5_000 and abs(n_valid - b["sums"][29]) <= 2  # validity thresholds sit on the gradient norm
    scale = np.abs(a["sums"]).max()
    assert np.abs(a["sums"] - b["sums"]).max() < 1e-5 * scale


def test_colour_term_quad_kernel_matches_the_64_per_wave_kernel(tmp_path):
    """Registration with the colour term: the four-lanes-per-query kernel with two split-fp16 images (default for
    decoders of up to two layers) against the 64-queries-per-wave fp32 kernel that deeper colour decoders still take
    (PIN_MLP=f32 routes there): all 32 Gauss-Newton sums, photometric and consistency-weight mode."""
    outs = {}
    for name, env in (("quad", {"PIN_MLP": "h2"}), ("wave", {"PIN_MLP": "f32"})):
        out = str(tmp_path / (name + ".npz"))
        e = dict(os.environ)
        e.update(env)
        subprocess.run([sys.executable, os.path.join(HERE, "_variant_worker.py"), out, "color"], check=True, env=e, timeout=600)
        outs[name] = np.load(out)
    for tag in ("photo", "consist"):
        a, b = outs["quad"][tag], outs["wave"][tag]
        assert abs(a[29] - b[29]) <= 2 and a[29] > 100
        assert np.abs(a - b).max() < 2e-5 * np.abs(b).max(), (tag, np.abs(a - b).max() / np.abs(b).max())
