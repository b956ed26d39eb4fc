"""The data-parallel mapper on the GPU: engine.MapTrainer with rank / world / communicator, end to end.

Only one GPU is leased to the tests and RCCL refuses two ranks on one device, so:
  * two ranks SHARING cuda:0 run the whole product path (shard evaluation by the HIP kernels, flat gradient
    exchange, dense Adam on both, certainty / ts merge) with the exchange carried by gloo through pinned host
    buffers (collective.HostStagedComm) -- and must reproduce the REFERENCE's whole-batch run on the fixture;
  * the RCCL transport itself (collective.RcclComm: pin_comm_*, pin_allreduce_grads, pin_dp_sync_side_effects
    through the C ABI on the caller's stream) runs with a communicator of one rank through the same code path."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from tests import golden_util as G

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _launch(tmp_path, world, transport, case):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    outs = [str(tmp_path / f"{transport}_{r}.npz") for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_dp_worker.py"), str(r), str(world), str(port), case,
                               transport, outs[r]]) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    return [np.load(o) for o in outs]


def _against_reference(d, r, grads=None):
    """Post-step state of a run against the reference's Mapper.mapping(2) on the same batches.  grads: the run whose
    recorded per-iteration gradients are compared (default: r itself; the lazy single-rank optimiser keeps the pending
    gradient of a row in the buffer until the row is read again, so its buffer is not a per-iteration gradient)."""
    g = r if grads is None else grads
    if grads is None:
        for it in range(2):
            gf, gd = d[f"map_gfeat{it}"], d[f"map_gdec{it}"]
            assert np.max(np.abs(g[f"gfeat{it}"].reshape(gf.shape) - gf)) < 1e-4 * np.abs(gf).max()
            assert np.max(np.abs(g[f"gdec{it}"] - gd)) < 1e-4 * np.abs(gd).max()
    G.adam_outliers(r["feats"], d["map_feat_after"], [g["gfeat0"], g["gfeat1"]], [d["map_gfeat0"], d["map_gfeat1"]], d["map_lr"])
    G.adam_outliers(r["dec"], d["map_dec_after"], [g["gdec0"], g["gdec1"]], [d["map_gdec0"], d["map_gdec1"]], d["map_lr"])
    np.testing.assert_allclose(r["cert"], d["map_cert_after"], rtol=1e-4, atol=1e-5)
    assert np.array_equal(r["tsu"], d["map_ts_after"])


@pytest.mark.parametrize("case", ["c2_wf", "c3_bigtable"])
def test_two_ranks_reproduce_the_reference(tmp_path, case):
    d = G.load(case)
    r0, r1 = _launch(tmp_path, 2, "host", case)
    assert str(r0["kind"]) == "host-staged gloo"
    # both ranks hold the identical model after the call (same reduced gradients -> same Adam)
    for key in ("feats", "dec", "cert", "tsu", "gfeat0", "gdec0", "gfeat1", "gdec1"):
        assert np.array_equal(r0[key].view(np.uint8), r1[key].view(np.uint8)), key
    _against_reference(d, r0)
    # and the single-rank product path (lazy exact Adam) lands on the same parameters
    (one,) = _launch(tmp_path, 1, "none", case)
    _against_reference(d, one, grads=r0)
    clean = (np.abs(d["map_gfeat0"]) > 1e-4 * np.abs(d["map_gfeat0"]).max()) & \
            (np.abs(d["map_gfeat1"]) > 1e-4 * np.abs(d["map_gfeat1"]).max())
    assert np.abs(one["feats"] - r0["feats"])[clean].max() < 1e-4


def test_rccl_transport_single_rank(tmp_path):
    """RCCL through the C ABI (dlopen, ncclGetUniqueId, ncclCommInitRank, in-place ncclAllReduce on the stream, the grouped
    certainty / ts exchange): with one rank the reductions are identities and the run must equal the reference."""
    d = G.load("c2_wf")
    (r,) = _launch(tmp_path, 1, "rccl", "c2_wf")
    assert str(r["kind"]) == "rccl"
    _against_reference(d, r)
