"""Load tests/golden/*.npz fixtures (written by oracle/make_golden.py from the real reference)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["c2_wf", "kitti_nwf", "c3_bigtable"]


def load(case):
    d = dict(np.load(os.path.join(GOLDEN, case + ".npz")))
    for k, v in list(d.items()):
        if v.shape == ():
            d[k] = v.item()
    return d


def dense_table(d, dtype=np.int64):
    t = np.full(int(d["buffer_size"]), -1, dtype)
    t[d["table_slots"]] = d["table_vals"]
    return t


def canon_knn(d2, idx, k):
    """Canonical (d2, candidate order) top-k of a reference [N,Kc] search result."""
    d2 = np.where(idx == -1, np.float32(9e3), d2)
    order = np.argsort(d2, axis=1, kind="stable")[:, :k]
    return np.take_along_axis(d2, order, 1), np.take_along_axis(idx, order, 1)
