"""The brick cache must return bit-identical kNN records to the direct hash-probe kernel
(which is itself bit-exact against the reference): collision-heavy small tables, the
non-local quirk, queries far from the map (fallback path), and a large synthetic map."""
import dataclasses

import numpy as np
import pytest
import torch

from oracle import pin_oracle as O
from tests import golden_util as G

pytestmark = pytest.mark.gpu


def _same(st, q, k, bricks, **kw):
    from pin_slam_amd import ops
    a = ops.knn_query(st, q, k, **kw)
    b = ops.knn_query(st, q, k, bricks=bricks, **kw)
    assert torch.equal(a[1], b[1]), "nn_count differs"
    assert torch.equal(a[0].view(torch.int32), b[0].view(torch.int32)), "kNN record differs"
    if kw.get("pose") is not None:
        assert torch.equal(a[2], b[2])
    return a


@pytest.mark.parametrize("case", G.CASES)
def test_bricks_equal_direct_probe_on_fixtures(case):
    from pin_slam_amd import ops
    from tests import gpu_util as U
    d = G.load(case)
    st = U.search_state(d)
    k = int(d["query_nn_k"])
    bricks = ops.BrickCache(d["neighbor_dx"], int(d["num_nei_cells"])).build(st, wait=True)
    assert bricks.n_bricks > 100 and bricks.n_entries > 1000
    rng = np.random.default_rng(0)
    far = rng.uniform(-200, 200, (500, 3)).astype(np.float32)          # mostly outside the cached bricks
    near = (d["local_neural_points"][::3] + rng.normal(0, 0.3, d["local_neural_points"][::3].shape)).astype(np.float32)
    for pts in (d["query"], d["reg_src"], d["map_coord0"], far, near):
        nbr, nn, _ = _same(st, U.dev(pts), k, bricks)
    assert int((nn > 0).sum()) > 100
    _same(st, U.dev(d["reg_src"]), k, bricks, pose=d["reg_Tinit"])


def test_bricks_reproduce_nonlocal_quirk():
    from pin_slam_amd import ops
    from tests import gpu_util as U
    d = G.load("c2_wf")
    mask, g2l = O.local_map_mask(d["neural_points"], d["point_ts_create"], [16.0, 0, 0], 6.0,
                                 travel_dist=d["travel_dist"], cur_ts=int(d["cur_ts"]),
                                 diff_travel_dist_local=d["diff_travel_dist_local"], reboot_ts=0)
    st = dataclasses.replace(U.search_state(d), global2local=U.dev(U.g2l_to_device_format(g2l, np.append(mask, True))))
    bricks = ops.BrickCache(d["neighbor_dx"], 2).build(st, wait=True)
    nbr, nn, _ = _same(st, U.dev(d["query"]), 8, bricks)
    _, idx, flag = U.nbr_split(nbr)
    assert flag.sum() > 50


def test_bricks_mode_is_checked_and_global_queries_use_direct_path():
    from pin_slam_amd import ops
    from tests import gpu_util as U
    d = G.load("c2_wf")
    st = U.search_state(d)
    bricks = ops.BrickCache(d["neighbor_dx"], 2).build(st, wait=True)
    with pytest.raises(RuntimeError):
        ops.knn_query(st, U.dev(d["query"]), 8, time_filtering=False, local=False, bricks=bricks)
    b2 = ops.BrickCache(d["neighbor_dx"], 2).build(st, time_filtering=False, local=False)
    _same(st, U.dev(d["query"]), 8, b2, time_filtering=False, local=False)


def test_bricks_large_synthetic_map():
    from pin_slam_amd import ops, synth
    from tests import gpu_util as U
    m = synth.build_map(layers=3, radius=40.0, raw_per_layer=400_000)
    P = len(m.positions)
    pos = U.dev(m.positions)
    pos4 = torch.empty((P, 4), dtype=torch.float32, device="cuda")
    ops.pack_positions(pos, torch.zeros(P, dtype=torch.int32, device="cuda"), pos4)
    dx, mv = ops.search_neighborhood(2, 0.5, 0.4)
    g2l = torch.arange(P + 1, dtype=torch.int32, device="cuda"); g2l[-1] = -1
    st = ops.SearchState(table=U.dev(m.table), pos4=pos4, cand_off=U.dev(ops.candidate_offsets(dx, m.buffer_size)),
                         n_points=P, resolution=0.4, max_valid_dist2=mv,
                         travel_dist=torch.zeros(1, device="cuda"), cur_ts=0, diff_travel_dist_local=400.0,
                         global2local=g2l)
    bricks = ops.BrickCache(dx, 2).build(st, wait=True)
    scan = synth.make_scan(m, n=50_000)
    pool, _ = synth.make_pool(m, n=50_000, sigma=0.6)
    for pts in (scan, pool):
        nbr, nn, _ = _same(st, U.dev(pts), 8, bricks)
    assert float(nn.float().mean()) > 20


@pytest.mark.parametrize("case,scale", [("c2_wf", 1.0), ("kitti_nwf", 1.0), ("c3_bigtable", 1.0), ("c2_wf", 30.0)])
def test_coherent_search_equals_full_search(case, scale):
    """pin_gn_knn_coherent over a Gauss-Newton-like sequence of poses (steps that shrink from centimetres to micrometres,
    `scale` x larger in the last case so that queries keep leaving their margins): every iteration's records and counts
    are the bits of a full search under the same pose, and once the steps are small most queries take the coherent path
    (their stored position stops following the pose)."""
    import ctypes as C
    from pin_slam_amd import _lib, ops
    from tests import gpu_util as U
    d = G.load(case)
    st = U.search_state(d)
    k = int(d["query_nn_k"])
    bricks = ops.BrickCache(d["neighbor_dx"], int(d["num_nei_cells"])).build(st, wait=True)
    rng = np.random.default_rng(4)
    src_np = np.concatenate([d["reg_src"], d["query"],
                             (d["local_neural_points"][::2] + rng.normal(0, 0.2, d["local_neural_points"][::2].shape))]).astype(np.float32)
    src = U.dev(src_np)
    n = src.shape[0]
    L = _lib.lib()
    sp, bc = st.params(time_filtering=True, local=True), bricks.params()
    nbr = torch.empty((n, k, 4), dtype=torch.float32, device="cuda")
    nn = torch.empty((n,), dtype=torch.int32, device="cuda")
    cur = torch.empty((n, 3), dtype=torch.float32, device="cuda")
    cst = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
    cwin = torch.zeros((n, 8), dtype=torch.int32, device="cuda")
    state = torch.zeros(_lib.PIN_GN_STATE_DOUBLES, dtype=torch.float64, device="cuda")
    T = np.eye(4)
    coherent_share = []
    for it in range(14):
        step = scale * 0.05 * 0.35 ** it  # 5 cm, 1.7 cm, 6 mm, ... , ~1e-7 m
        ang = step * 0.02
        dT = np.eye(4)
        dT[:3, :3] = [[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]]
        dT[:3, 3] = step * np.array([0.6, -0.5, 0.3])
        T = dT @ T
        state[:16] = torch.from_numpy(T.reshape(-1)).cuda()
        _lib.check(L.pin_gn_knn_coherent(C.byref(sp), C.byref(bc), src.data_ptr(), n, k, state.data_ptr(), cur.data_ptr(),
                                         nbr.data_ptr(), nn.data_ptr(), cst.data_ptr(), cwin.data_ptr(), it,
                                         ops._stream()), "pin_gn_knn_coherent")
        ref_nbr, ref_nn, ref_cur = ops.knn_query(st, src, k, pose=T, bricks=bricks)
        assert torch.equal(cur, ref_cur)
        assert torch.equal(nn, ref_nn), f"nn_count differs at iteration {it}"
        assert torch.equal(nbr.view(torch.int32), ref_nbr.view(torch.int32)), f"kNN record differs at iteration {it}"
        coherent_share.append(float((cst[:, :3] != cur).any(1).float().mean().item()))
    assert coherent_share[0] == 0.0
    if scale == 1.0:
        assert coherent_share[-1] > 0.8, coherent_share
    else:
        assert 0.0 < max(coherent_share) and min(coherent_share[1:]) < 0.9, coherent_share


@pytest.mark.parametrize("case,scale", [("c2_wf", 1.0), ("kitti_nwf", 1.0), ("c3_bigtable", 1.0), ("c2_wf", 30.0)])
def test_listed_search_equals_full_search(case, scale):
    """pin_gn_knn_listed (candidate lists kept across the iterations of a registration; what Tracker.tracking launches): over a
    Gauss-Newton-like sequence of poses -- `scale` 30: steps of up to 1.5 m, so queries change voxel all the time and lists are
    rebuilt mid-sequence -- every iteration's records, counts and transformed points are the bits of a full search under the
    same pose; the lists are used (the voxel stored with a list follows the query only when it changes voxel)."""
    import ctypes as C
    from pin_slam_amd import _lib, ops
    from tests import gpu_util as U
    d = G.load(case)
    st = U.search_state(d)
    k = int(d["query_nn_k"])
    bricks = ops.BrickCache(d["neighbor_dx"], int(d["num_nei_cells"])).build(st, wait=True)
    rng = np.random.default_rng(4)
    far_out = np.array([[1e4, 0, 0], [0, -1e4, 3.0]], np.float32)  # nothing near: empty lists
    src_np = np.concatenate([d["reg_src"], d["query"], far_out,
                             (d["local_neural_points"][::2] + rng.normal(0, 0.2, d["local_neural_points"][::2].shape))]).astype(np.float32)
    src = U.dev(src_np)
    n = src.shape[0]
    L = _lib.lib()
    sp, bc = st.params(time_filtering=True, local=True), bricks.params()
    stride = int(L.pin_knn_list_stride(int(sp.n_cand)))
    assert stride >= int(sp.n_cand)
    nbr = torch.empty((n, k, 4), dtype=torch.float32, device="cuda")
    nn = torch.empty((n,), dtype=torch.int32, device="cuda")
    cur = torch.empty((n, 3), dtype=torch.float32, device="cuda")
    cells = torch.full((n, 4), 123456, dtype=torch.int32, device="cuda")
    lists = torch.zeros((n, stride), dtype=torch.int32, device="cuda")
    state = torch.zeros(_lib.PIN_GN_STATE_DOUBLES, dtype=torch.float64, device="cuda")
    T = np.eye(4)
    moved = []
    for it in range(14):
        step = scale * 0.05 * 0.35 ** it
        ang = step * 0.02
        dT = np.eye(4)
        dT[:3, :3] = [[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]]
        dT[:3, 3] = step * np.array([0.6, -0.5, 0.3])
        T = dT @ T
        state[:16] = torch.from_numpy(T.reshape(-1)).cuda()
        before = cells.clone()
        _lib.check(L.pin_gn_knn_listed(C.byref(sp), C.byref(bc), src.data_ptr(), n, k, state.data_ptr(), cur.data_ptr(),
                                       nbr.data_ptr(), nn.data_ptr(), cells.data_ptr(), lists.data_ptr(), int(it == 0),
                                       ops._stream()), "pin_gn_knn_listed")
        ref_nbr, ref_nn, ref_cur = ops.knn_query(st, src, k, pose=T, bricks=bricks)
        assert torch.equal(cur, ref_cur)
        assert torch.equal(nn, ref_nn), f"nn_count differs at iteration {it}"
        assert torch.equal(nbr.view(torch.int32), ref_nbr.view(torch.int32)), f"kNN record differs at iteration {it}"
        # the stored voxel is the query's voxel; the stored length counts the occupied candidate cells (>= the accepted ones)
        vox = torch.floor(cur / np.float32(d["resolution"])).to(torch.int32)
        assert torch.equal(cells[:, :3], vox)
        has = cells[:, 3] >= 0
        assert bool((cells[has, 3] >= nn[has]).all()) and bool((cells[:, 3] >= -1).all())
        moved.append(float((before[:, :3] != cells[:, :3]).any(1).float().mean().item()))
    assert moved[0] == 1.0
    if scale == 1.0:
        assert max(moved[4:]) < 0.01, moved  # millimetre steps: (almost) nobody changes voxel, every search runs off its list
    else:
        assert moved[1] > 0.5 and moved[-1] < 0.01, moved


def _cache_contents(b):
    """{brick key: (mask, {cell bit: entry bits})} of a built cache (ids and entry ranges differ between the two builds)."""
    nb = b.n_bricks
    keys = b.brick_keys[:nb].cpu().numpy()
    masks = b.brick_mask[:nb].cpu().numpy().astype(np.uint64)
    bases = b.brick_base[:nb].cpu().numpy()
    ent = b.entries.cpu().numpy().view(np.uint32)
    out = {}
    for key, m, base in zip(keys.tolist(), masks.tolist(), bases.tolist()):
        cells = {}
        e = base
        for bit in range(64):
            if (m >> bit) & 1:
                cells[bit] = tuple(ent[e].tolist()) if base >= 0 else None
                e += 1
        out[key] = (m, cells)
    return out


@pytest.mark.parametrize("case", ["c2_wf", "kitti_nwf", "c3_bigtable", "nonlocal"])
def test_point_driven_build_equals_cell_driven_build(case):
    """pin_brick_build from the POINTS (r04: own bricks + mask-based dilation, one table probe per point) against the build
    from the CELLS of every brick: the same set of bricks, the same occupancy masks, the same entry per cell, bit for bit --
    on fixtures with a collision-heavy table, per-neighbour settings and the non-local quirk (points outside the local map
    still get their entries)."""
    from pin_slam_amd import ops
    from tests import gpu_util as U
    d = G.load("c2_wf" if case == "nonlocal" else case)
    st = U.search_state(d)
    if case == "nonlocal":
        mask, g2l = O.local_map_mask(d["neural_points"], d["point_ts_create"], [16.0, 0, 0], 6.0, travel_dist=d["travel_dist"],
                                     cur_ts=int(d["cur_ts"]), diff_travel_dist_local=d["diff_travel_dist_local"], reboot_ts=0)
        st = dataclasses.replace(st, global2local=U.dev(U.g2l_to_device_format(g2l, np.append(mask, True))))
    n = int(d["num_nei_cells"])
    a = ops.BrickCache(d["neighbor_dx"], n)
    a.by_points = True
    a.build(st, wait=True)
    b = ops.BrickCache(d["neighbor_dx"], n)
    b.by_points = False
    b.build(st, wait=True)
    assert a.build_ws is not None and b.build_ws is None
    assert a.n_bricks == b.n_bricks and a.n_entries == b.n_entries and a.n_bricks > 100
    ca, cb = _cache_contents(a), _cache_contents(b)
    assert ca.keys() == cb.keys()
    for key in ca:
        assert ca[key] == cb[key], f"brick {key:#x} differs"
    q = U.dev(d["query"])
    ra, rb = ops.knn_query(st, q, int(d["query_nn_k"]), bricks=a), ops.knn_query(st, q, int(d["query_nn_k"]), bricks=b)
    assert torch.equal(ra[0].view(torch.int32), rb[0].view(torch.int32)) and torch.equal(ra[1], rb[1])


@pytest.mark.parametrize("case", ["c2_wf", "c3_bigtable", "bench_map"])
def test_narrow_build_equals_full_width_build(case):
    """pin_brick_cache.build_grid (r05): launches of the build at most that many blocks wide, each block walking its share of
    the points / bricks / directory slots -- the per-frame build beside the caller's small launches.  Whatever the width, the
    cache holds the same bricks, the same occupancy masks and the same entry per cell, and a search through it returns the
    same records; `bench_map` = the 2.2 M-point map of the benchmark (8 700 units of work walked by 3 / 64 / 512 blocks)."""
    from pin_slam_amd import ops
    from tests import gpu_util as U
    if case == "bench_map":
        from pin_slam_amd import synth
        m = synth.build_map(layers=16)
        P = len(m.positions)
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
        pos4 = torch.empty((P, 4), dtype=torch.float32, device="cuda")
        ops.pack_positions(dev(m.positions), torch.zeros(P, dtype=torch.int32, device="cuda"), pos4)
        dx, mv = ops.search_neighborhood(2, 0.5, 0.4)
        g2l = torch.arange(P + 1, dtype=torch.int32, device="cuda")
        g2l[-1] = -1
        st = ops.SearchState(table=dev(m.table), pos4=pos4, cand_off=dev(ops.candidate_offsets(dx, m.buffer_size)), n_points=P,
                             resolution=0.4, max_valid_dist2=mv, travel_dist=torch.zeros(1, device="cuda"), cur_ts=0,
                             diff_travel_dist_local=410.0, global2local=g2l)
        n, k = 2, 8
        q = dev(synth.make_scan(m, n=20_000, seed=5))
        widths = (3, 512)
    else:
        d = G.load(case)
        st = U.search_state(d)
        dx, n, k = d["neighbor_dx"], int(d["num_nei_cells"]), int(d["query_nn_k"])
        q = U.dev(d["query"])
        widths = (1, 7, 64)
    full = ops.BrickCache(dx, n)
    full.build(st, wait=True)
    assert full.build_grid == 0 and full.n_bricks > 100
    ref_contents = _cache_contents(full) if case != "bench_map" else None
    ref = ops.knn_query(st, q, k, bricks=full)
    for w in widths:
        nb = ops.BrickCache(dx, n)
        nb.build_grid = w
        nb.build(st, wait=True)
        assert nb.params().build_grid == w
        assert (nb.n_bricks, nb.n_entries) == (full.n_bricks, full.n_entries), w
        if ref_contents is not None:
            got = _cache_contents(nb)
            assert got.keys() == ref_contents.keys()
            for key in got:
                assert got[key] == ref_contents[key], f"width {w}: brick {key:#x} differs"
        r = ops.knn_query(st, q, k, bricks=nb)
        assert torch.equal(r[0].view(torch.int32), ref[0].view(torch.int32)) and torch.equal(r[1], ref[1]), w
