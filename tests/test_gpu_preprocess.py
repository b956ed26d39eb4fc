"""GPU parity of the preprocess_frame data path (crop_frame, intrinsic_correct, deskewing and the
two voxel down-sampling passes) through the C ABI against the reference fixture and the oracle."""
import numpy as np
import pytest
import torch

from oracle import pin_oracle as O
from tests import golden_util as G

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pp():
    return G.load("preprocess")


def test_preprocess_steps_follow_reference(pp):
    from pin_slam_amd import preprocess as PP
    d = pp
    scan, ts = torch.from_numpy(d["scan"]).cuda(), torch.from_numpy(d["ts"]).cuda()
    idx = PP.voxel_down_sample_torch(scan[:, :3], d["vox_down_m"])
    assert idx.dtype == torch.int64 and np.array_equal(idx.cpu().numpy(), d["idx_train"])
    pc, pts_ts = PP.crop_frame(PP.gather(scan, idx), ts[idx], d["min_z"], d["max_z"], d["min_range"], d["max_range"])
    assert np.array_equal(pc.cpu().numpy(), d["cropped"]) and np.array_equal(pts_ts.cpu().numpy(), d["cropped_ts"])
    pc = PP.intrinsic_correct(pc.clone(), d["correct_deg"])
    np.testing.assert_allclose(pc.cpu().numpy(), d["corrected"], rtol=1e-6, atol=1e-6)
    ref = torch.from_numpy(d["corrected"]).cuda()  # continue from the reference's bits: voxel ids are discontinuous
    idx2 = PP.voxel_down_sample_torch(ref[:, :3], d["source_vox_down_m"])
    assert np.array_equal(idx2.cpu().numpy(), d["idx_source"])
    src = PP.gather(ref, idx2)[:, :3].contiguous()
    assert np.array_equal(src.cpu().numpy(), d["source"])
    out = PP.deskewing(src.clone(), torch.from_numpy(d["source_ts"]).cuda(), torch.from_numpy(d["last_odom_tran"]))
    np.testing.assert_allclose(out.cpu().numpy(), d["deskewed"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(out.cpu().numpy(), O.deskewing(d["source"], d["source_ts"], d["last_odom_tran"]), rtol=0, atol=2e-5)


def test_scan_preprocessor_matches_oracle_chain():
    """100k-point scan through ScanPreprocessor vs the oracle functions chained the same way."""
    from pin_slam_amd import preprocess as PP
    from pin_slam_amd.config import PinConfig
    cfg = PinConfig(vox_down_m=0.08, source_vox_down_m=0.8, min_range=2.5, max_range=60.0, min_z=-5.0, max_z=60.0, deskew=True)
    g = torch.Generator().manual_seed(2)
    n = 100_000
    r = 70.0 * torch.sqrt(torch.rand(n, generator=g)); th = 6.2831853 * torch.rand(n, generator=g)
    scan = torch.stack([r * torch.cos(th), r * torch.sin(th), -2 + 0.3 * torch.sin(0.5 * r) + 0.5 * torch.randn(n, generator=g),
                        torch.rand(n, generator=g)], 1).float()
    ts = torch.rand(n, generator=g).float()
    T = np.eye(4); T[:3, 3] = [0.9, 0.02, 0.0]
    c, s = np.cos(0.03), np.sin(0.03); T[:3, :3] = [[c, -s, 0], [s, c, 0], [0, 0, 1]]
    pc, pts_ts, src, _ = PP.ScanPreprocessor(cfg)(scan.cuda(), ts.cuda(), last_odom_tran=T, frame_id=3)
    sn, tn = scan.numpy(), ts.numpy()
    i1 = O.voxel_down_sample(sn[:, :3], cfg.vox_down_m)
    m = O.crop_frame_mask(sn[i1], cfg.min_z, cfg.max_z, cfg.min_range, cfg.max_range)
    ref_pc, ref_ts = sn[i1][m], tn[i1][m]
    assert np.array_equal(pc.cpu().numpy(), ref_pc) and np.array_equal(pts_ts.cpu().numpy(), ref_ts)
    i2 = O.voxel_down_sample(ref_pc[:, :3], cfg.source_vox_down_m)
    ref_src = O.deskewing(ref_pc[i2][:, :3], ref_ts[i2], T)
    np.testing.assert_allclose(src.cpu().numpy(), ref_src, rtol=0, atol=3e-5)
    assert 1000 < len(i2) < len(i1)


@pytest.mark.parametrize("deskew,correct", [(True, 0.0), (False, 0.205), (True, 0.205)])
def test_fused_chain_equals_the_stages(monkeypatch, deskew, correct):
    """pin_preprocess_frame (one call, stage counts on the device, one read-back) against the stage-by-stage path of the same
    class: identical clouds, timestamps and registration source, bit for bit (incl. the KITTI correction in between, which moves
    points across voxel borders of the second down-sampling)."""
    from pin_slam_amd import preprocess as PP
    from pin_slam_amd.config import PinConfig
    cfg = PinConfig(vox_down_m=0.08, source_vox_down_m=0.8, min_range=2.5, max_range=60.0, min_z=-5.0, max_z=60.0, deskew=deskew,
                    kitti_correction_on=correct != 0.0, correction_deg=correct)
    g = torch.Generator().manual_seed(5)
    n = 120_000
    r = 70.0 * torch.sqrt(torch.rand(n, generator=g)); th = 6.2831853 * torch.rand(n, generator=g)
    scan = torch.stack([r * torch.cos(th), r * torch.sin(th), -2 + 0.3 * torch.sin(0.5 * r) + 0.5 * torch.randn(n, generator=g),
                        torch.rand(n, generator=g)], 1).float().cuda()
    ts = torch.rand(n, generator=g).float().cuda()
    T = np.eye(4); T[:3, 3] = [0.9, 0.02, 0.0]
    c, s = np.cos(0.03), np.sin(0.03); T[:3, :3] = [[c, -s, 0], [s, c, 0], [0, 0, 1]]
    fused = PP.ScanPreprocessor(cfg)
    assert fused.fused
    staged = PP.ScanPreprocessor(cfg)
    staged.fused = False
    for frame_id in (0, 3):
        a = fused(scan, ts, last_odom_tran=T, frame_id=frame_id)
        b = staged(scan, ts, last_odom_tran=T, frame_id=frame_id)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and a[0].shape[0] > 50_000
        if frame_id == 0:
            assert a[2] is None and b[2] is None
        else:
            assert torch.equal(a[2], b[2]) and 1000 < a[2].shape[0] < a[0].shape[0]
    # no timestamps: no deskewing, no ts output
    a, b = fused(scan, None, frame_id=2), staged(scan, None, frame_id=2)
    assert a[1] is None and torch.equal(a[0], b[0]) and torch.equal(a[2], b[2])


@pytest.mark.parametrize("fused", [True, False])
def test_chain_on_its_own_stream_does_not_wait_for_the_main_stream(fused):
    """ScanPreprocessor(..., stream=s): the same bits as on the current stream, results usable on the current stream when the
    call returns, and the call does not queue behind the main stream: with ~40 ms of work pending there it returns long
    before that work ends (the pipeline bench.py times: the scan chain of frame f+1 beside Mapper.mapping of frame f)."""
    import time
    from pin_slam_amd import preprocess as PP
    from pin_slam_amd.config import PinConfig
    cfg = PinConfig(vox_down_m=0.08, source_vox_down_m=0.8, min_range=2.5, max_range=60.0, min_z=-5.0, max_z=60.0, deskew=True)
    g = torch.Generator().manual_seed(11)
    n = 100_000
    r = 70.0 * torch.sqrt(torch.rand(n, generator=g)); th = 6.2831853 * torch.rand(n, generator=g)
    scan = torch.stack([r * torch.cos(th), r * torch.sin(th), -2 + 0.5 * torch.randn(n, generator=g), torch.rand(n, generator=g)], 1).float().cuda()
    ts = torch.rand(n, generator=g).float().cuda()
    T = np.eye(4); T[:3, 3] = [0.9, 0.02, 0.0]
    plain, other = PP.ScanPreprocessor(cfg), PP.ScanPreprocessor(cfg)
    plain.fused = other.fused = fused
    want = plain(scan, ts, last_odom_tran=T, frame_id=3)
    side = torch.cuda.Stream()
    other(scan, ts, last_odom_tran=T, frame_id=3, stream=side)  # (first call: allocations)
    a = torch.randn(6144, 6144, device="cuda")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        a = a @ a * 1e-3
    torch.cuda.synchronize()
    busy = time.perf_counter() - t0
    for _ in range(8):  # pending on the main stream while the chain runs on its own
        a = a @ a * 1e-3
    t0 = time.perf_counter()
    got = other(scan, ts, last_odom_tran=T, frame_id=3, stream=side)
    t_call = time.perf_counter() - t0
    main_done_at_return = torch.cuda.current_stream().query()
    s = got[0].sum() + got[2].sum()  # used on the main stream at once, no event
    torch.cuda.synchronize()
    for w, x in zip(want, got):
        assert (w is None and x is None) or torch.equal(w, x)
    assert torch.isfinite(s)
    if busy > 0.01:  # (the matmuls were long enough to tell)
        assert not main_done_at_return and t_call < 0.6 * busy, (t_call, busy)


def test_begin_finish_is_the_call_in_two_halves():
    """ScanPreprocessor.begin / finish: the chain queued on a stream at once (here from a loader THREAD, as bench.py does), its
    results collected later on the main thread -- the bits of the plain call; one ticket at a time; a configuration whose set-up
    needs a read-back runs whole in finish()."""
    import threading
    from pin_slam_amd import preprocess as PP
    from pin_slam_amd.config import PinConfig
    g = torch.Generator().manual_seed(13)
    n = 80_000
    r = 70.0 * torch.sqrt(torch.rand(n, generator=g)); th = 6.2831853 * torch.rand(n, generator=g)
    scan = torch.stack([r * torch.cos(th), r * torch.sin(th), -2 + 0.5 * torch.randn(n, generator=g), torch.rand(n, generator=g)], 1).float().cuda()
    ts = torch.rand(n, generator=g).float().cuda()
    T = np.eye(4); T[:3, 3] = [0.9, 0.02, 0.0]
    side = torch.cuda.Stream()
    for adaptive in (False, True):
        cfg = PinConfig(vox_down_m=0.08, source_vox_down_m=0.8, min_range=2.5, max_range=60.0, min_z=-5.0, max_z=60.0, deskew=True,
                        adaptive_range_on=adaptive)
        plain, two = PP.ScanPreprocessor(cfg), PP.ScanPreprocessor(cfg)
        want = plain(scan, ts, last_odom_tran=T, frame_id=3)
        box = {}
        th_ = threading.Thread(target=lambda: box.update(tk=two.begin(scan, ts, last_odom_tran=T, frame_id=3, stream=side)))
        th_.start()
        a = torch.randn(2048, 2048, device="cuda")
        a = a @ a  # (the main thread is free meanwhile)
        th_.join()
        tk = box["tk"]
        assert ("st" in tk) == (not adaptive)
        with pytest.raises(RuntimeError):
            two.begin(scan, ts, last_odom_tran=T, frame_id=4, stream=side)
        with pytest.raises(RuntimeError):
            two(scan, ts, last_odom_tran=T, frame_id=4)
        got = two.finish(tk)
        s = got[0].sum() + got[2].sum()
        torch.cuda.synchronize()
        for w, x in zip(want, got):
            assert (w is None and x is None) or torch.equal(w, x)
        assert torch.isfinite(s)
        got2 = two.finish(two.begin(scan, ts, last_odom_tran=T, frame_id=3, stream=side))  # (and again: the ticket was closed)
        assert torch.equal(want[0], got2[0])
