"""The normal-equation solve of the registration loop (csrc/gn_solve.h: one wave, an element of the 6x7 system per lane) and the two
places it runs in: a launch of its own (pin_gn_solve) and the last block of the tile kernel (pin_gn_accumulate_solve on a state from
pin_gn_loop_init).  Reference: implicit_reg and the bookkeeping of Tracker.tracking (utils/tracker.py:656-679, 147-184), restated in
numpy below and in oracle.pin_oracle."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import pin_oracle as O
from tests import golden_util as G

pytestmark = pytest.mark.gpu


def _pack_sums(N, b_neg, w_sum, abs_res_sum, cnt, wrr, rng):
    """double[REPLICAS][NSUMS] whose replica sum is the packed system: [0..20] upper triangle of N, [21..26] J^T W r, 27 sum w,
    28 sum |r|, 29 count, 30 sum w r^2 -- spread over the replicas at random."""
    from pin_slam_amd import _lib
    s = np.zeros(_lib.PIN_GN_NSUMS)
    o = 0
    for a in range(6):
        for b in range(a, 6):
            s[o] = N[a, b]
            o += 1
    s[21:27] = b_neg
    s[27], s[28], s[29], s[30] = w_sum, abs_res_sum, cnt, wrr
    parts = rng.dirichlet(np.ones(_lib.PIN_GN_REPLICAS), size=_lib.PIN_GN_NSUMS).T  # [R, NSUMS], columns sum to 1
    rep = parts * s[None, :]
    rep[-1] += s - rep.sum(0)
    return np.ascontiguousarray(rep), s  # (C order: the product above inherits the transposed layout of `parts`)


def _expected(s, T0, lm):
    cnt = round(s[29])
    scale = cnt / (2.0 * s[27])
    N = np.zeros((6, 6))
    o = 0
    for a in range(6):
        for b in range(a, 6):
            N[a, b] = N[b, a] = scale * s[o]
            o += 1
    g = -scale * s[21:27]
    t = np.linalg.solve(N + lm * np.diag(np.diag(N)), g)
    dT = np.eye(4)
    dT[:3, :3] = O.expmap(t[:3])
    dT[:3, 3] = t[3:]
    return dT @ T0, N, scale * s[30] / cnt, s[28] / cnt * 100.0


def _lp(iters=50, early_exit=False, lm=1e-4):
    from pin_slam_amd import _lib
    lp = _lib.GnLoopParams()
    lp.lm_lambda, lp.term_thre_deg, lp.term_thre_m = lm, 0.01, 0.001
    lp.min_valid_ratio, lp.max_increment_ratio, lp.min_valid_points = 0.2, 1.1, 30
    lp.iter_n, lp.early_exit = iters, int(early_exit)
    return lp


@pytest.mark.parametrize("case", ["well", "scaled", "big_rotation", "few_points", "done"])
def test_solve_kernel_against_numpy(case):
    """pin_gn_solve on synthetic sums: pose, N_raw, mse, residual, counters and the cleared sums against numpy's solve (float64)."""
    from pin_slam_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng({"well": 1, "scaled": 2, "big_rotation": 3, "few_points": 4, "done": 5}[case])
    J = rng.normal(size=(400, 6)) * (np.array([30, 30, 30, 1, 1, 1.0]) if case != "scaled" else np.array([300, 0.1, 40, 1, 50, 0.02]))
    w = rng.uniform(0.2, 1.0, size=400)
    N = (J * w[:, None]).T @ J
    r = rng.normal(size=400) * 0.02
    if case == "big_rotation":  # a step of 0.88 rad: beyond the range of the series in gn_solve.h
        r = r - J @ np.array([0.6, -0.5, 0.4, 0.3, -0.2, 0.1])
    b = (J * w[:, None]).T @ r
    cnt = 400 if case != "few_points" else 7
    rep, s = _pack_sums(N, b, w.sum(), np.abs(r).sum(), cnt, (w * r * r).sum(), rng)
    T0 = np.eye(4)
    T0[:3, :3] = O.expmap(np.array([0.2, -0.1, 0.4]))
    T0[:3, 3] = [3.0, -2.0, 0.5]
    lp = _lp()
    state = torch.zeros(_lib.PIN_GN_STATE_DOUBLES, dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    _lib.check(L.pin_gn_loop_init(state.data_ptr(), np.ascontiguousarray(T0).ctypes.data, 1000, C.byref(lp), stream), "pin_gn_loop_init")
    if case == "done":
        state[_lib_index("DONE")] = 1.0
    sums = torch.from_numpy(rep).cuda()
    _lib.check(L.pin_gn_solve(sums.data_ptr(), state.data_ptr(), C.byref(lp), stream), "pin_gn_solve")
    st = state.cpu().numpy()
    if case == "done":  # nothing moves, the sums stay
        assert np.array_equal(st[:16].reshape(4, 4), T0) and st[22] == 0 and np.array_equal(sums.cpu().numpy(), rep)
        return
    assert not sums.cpu().numpy().any()
    assert st[22] == 1 and st[18] == cnt
    if case == "few_points":  # tracker.py:430-432: no step; not enough points: the loop ends invalid
        assert np.array_equal(st[:16].reshape(4, 4), T0) and st[19] == 0 and st[21] == 1 and st[17] == 0
        return
    T, N_raw, mse, res_cm = _expected(s, T0, lp.lm_lambda)
    np.testing.assert_allclose(st[:16].reshape(4, 4), T, rtol=0, atol=(1e-7 if case == "scaled" else 1e-11) * max(1.0, np.abs(T).max()))
    if case == "big_rotation":
        assert np.arccos((np.trace((st[:16].reshape(4, 4) @ np.linalg.inv(T0))[:3, :3]) - 1) / 2) > 0.5
    np.testing.assert_allclose(st[24:60].reshape(6, 6), N_raw, rtol=1e-13, atol=0)
    np.testing.assert_allclose([st[23], st[17]], [mse, res_cm], rtol=1e-13)
    assert st[19] == 1 and st[21] == 0 and st[16] == st[17]
    R = st[:16].reshape(4, 4)[:3, :3]
    np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-12)  # (also beyond half a radian: the device library's sin / cos)


def _lib_index(name):
    return {"DONE": 21}[name]


@pytest.fixture(scope="module", params=G.CASES)
def gold(request):
    from tests import gpu_util as U
    d = G.load(request.param)
    d["table"] = G.dense_table(d)
    d["st"] = U.search_state(d, d["table"].astype(np.int32))
    d["fs_loc"] = U.field_state(d, local=True)
    return d


@pytest.mark.parametrize("early_exit", [True, False])
def test_both_forms_of_the_iteration_agree(gold, early_exit):
    """The same registration with the solve (a) in a launch of its own, (b) in the tile kernel's last block: pose, counts,
    residual, iterations and flags agree (the sums are float64 atomics of per-block float32 sums: their order differs between
    runs of ONE form as well, hence a tolerance, not bits) -- with the reference's stopping rules and with a fixed number of
    iterations."""
    from pin_slam_amd import engine, ops
    from tests import gpu_util as U
    from tests.test_gpu_parity import _gn_params
    d = gold
    src = U.dev(d["reg_src"])
    iters = int(d["cfg_reg_iter_n"]) if early_exit else 7
    out = {}
    for form in ("own_launch", "last_block"):
        gn = engine.GNTracker(d["st"], d["fs_loc"], _gn_params(d), d["cfg_reg_lm_lambda"], src.shape[0])
        gn.bricks = ops.BrickCache(d["neighbor_dx"], int(d["num_nei_cells"])).build(d["st"])
        gn.fuse_solve = form == "last_block"
        for _ in range(2):  # (twice: the second call starts from what the first one left in the sums and the ticket)
            out[form] = gn.track(src, d["reg_Tinit"], iters, term_deg=d["cfg_reg_term_thre_deg"], term_m=d["cfg_reg_term_thre_m"],
                                 early_exit=early_exit)
        assert not gn.sums.cpu().numpy().any() and gn.state_host.numpy()[62] == 0.0
    Ta, ca, ra, ia, va, xa = out["own_launch"]
    T, c, r, i, v, x = out["last_block"]
    np.testing.assert_allclose(T, Ta, rtol=0, atol=2e-7)
    assert (c, i, v, x["converged"]) == (ca, ia, va, xa["converged"])
    np.testing.assert_allclose([r, x["mse"]], [ra, xa["mse"]], rtol=1e-5)
    np.testing.assert_allclose(x["N_raw"], xa["N_raw"], rtol=1e-5, atol=1e-6 * np.abs(xa["N_raw"]).max())
    if early_exit:
        assert ia < iters and va == bool(d["trk_valid"])
        np.testing.assert_allclose(Ta[:3, 3], d["trk_T"][:3, 3], rtol=0, atol=1e-4)
    else:
        assert ia == iters


def test_presorted_registration_is_the_sorted_one(gold):
    """GNTracker.presort on another stream, then track() on the same tensor: the order is taken from the buffer (no sort launch in
    track), the result is the one of a call that sorts itself; a call on another tensor in between sorts as usual."""
    from pin_slam_amd import engine, ops
    from tests import gpu_util as U
    from tests.test_gpu_parity import _gn_params
    d = gold
    src = U.dev(d["reg_src"])
    kw = dict(term_deg=d["cfg_reg_term_thre_deg"], term_m=d["cfg_reg_term_thre_m"])
    gn = engine.GNTracker(d["st"], d["fs_loc"], _gn_params(d), d["cfg_reg_lm_lambda"], src.shape[0])
    gn.sort_min_points = 1
    gn.bricks = ops.BrickCache(d["neighbor_dx"], int(d["num_nei_cells"])).build(d["st"])
    want = gn.track(src, d["reg_Tinit"], int(d["cfg_reg_iter_n"]), **kw)
    side = torch.cuda.Stream()
    calls = []
    sort = gn._sort_into_buffer
    gn._sort_into_buffer = lambda s, stream: (calls.append(s.data_ptr()), sort(s, stream))[1]
    gn.presort(src, stream=side)
    assert calls == [src.data_ptr()] and gn._presorted is not None
    got = gn.track(src, d["reg_Tinit"], int(d["cfg_reg_iter_n"]), **kw)
    assert calls == [src.data_ptr()] and gn._presorted is None  # (track() did not sort again)
    np.testing.assert_allclose(got[0], want[0], rtol=0, atol=2e-7)
    assert got[1:5] == want[1:5] or (got[1], got[3], got[4]) == (want[1], want[3], want[4])
    other = src.clone()
    gn.presort(src, stream=side)
    got2 = gn.track(other, d["reg_Tinit"], int(d["cfg_reg_iter_n"]), **kw)  # another tensor: the stale order is not used
    assert calls[-1] == other.data_ptr() and len(calls) == 3
    np.testing.assert_allclose(got2[0], want[0], rtol=0, atol=2e-7)
