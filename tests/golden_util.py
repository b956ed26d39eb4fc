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


def adam_outliers(p, p_ref, grads, ref_grads, lr, tol=1e-4):
    """Post-Adam parameters against the reference's after len(grads) iterations.

    With eps = 1e-15 Adam turns every non-zero gradient into a step of about lr whatever its size
    (step 1 is lr * sign(g)), so a parameter whose gradient is at the rounding-noise level of the gradient
    accumulation can legitimately step the other way.  The noise level is MEASURED: `noise_it` = the largest
    difference between our gradient and the reference's in iteration `it`.  A relative gradient error eps moves
    an Adam step by about eps * lr, so entries whose reference gradient exceeds 100 x noise_it in every iteration
    (or is exactly zero in both) must agree to `tol`; the remaining, noise-dominated, entries are bounded by the
    largest possible total step.  Returns (fraction of noise-dominated entries, worst clean difference)."""
    p, p_ref = np.asarray(p, np.float64), np.asarray(p_ref, np.float64)
    clean = np.ones(p.shape, bool)
    for g, gr in zip(grads, ref_grads):
        g, gr = np.asarray(g, np.float64).reshape(p.shape), np.asarray(gr, np.float64).reshape(p.shape)
        noise = np.abs(g - gr).max()
        clean &= (np.abs(gr) >= 100.0 * noise) | ((gr == 0) & (g == 0))
    diff = np.abs(p - p_ref)
    worst_clean = float(diff[clean].max()) if clean.any() else 0.0
    assert worst_clean < tol, f"post-Adam parameters differ by {worst_clean} where the gradients are well above the noise"
    bound = 2.0 * lr * len(grads) * 1.05
    assert diff.max() <= bound, f"a parameter moved by more than Adam can move it: {diff.max()} > {bound}"
    return float(1.0 - clean.mean()), worst_clean
