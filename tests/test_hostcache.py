"""pin_slam_amd.hostcache: a host copy is handed out only for the very tensor contents it was recorded for."""
import numpy as np
import torch

from pin_slam_amd import hostcache


def test_lookup_follows_storage_and_version():
    a = np.arange(16, dtype=np.float64).reshape(4, 4)
    t = torch.tensor(a)
    assert hostcache.lookup(t) is None or not np.array_equal(hostcache.lookup(t), a + 1)
    hostcache.remember(t, a)
    got = hostcache.lookup(t)
    assert got is not None and np.array_equal(got, a)
    got[0, 0] = -1.0  # a copy: the cached array is not handed out itself
    assert hostcache.lookup(t)[0, 0] == 0.0
    d = t.detach()  # what SLAMDataset.update_odom_pose keeps (slam_dataset.py:514): same storage, same version counter
    assert np.array_equal(hostcache.lookup(d), a)
    assert hostcache.lookup(t[:3, 3]) is None  # another view of the storage: no entry of that shape
    assert hostcache.lookup(t.clone()) is None  # other storage
    t[0, 0] = 5.0  # an in-place write bumps the version counter of every view
    assert hostcache.lookup(t) is None and hostcache.lookup(d) is None
    assert np.array_equal(hostcache.to_host(t), t.numpy())  # re-read and remembered
    assert np.array_equal(hostcache.lookup(t), t.numpy())


def test_entries_keep_their_tensor_alive_and_are_bounded():
    ptrs = set()
    for i in range(10):
        t = torch.full((4, 4), float(i), dtype=torch.float64)
        hostcache.remember(t, t.numpy())
        ptrs.add(t.data_ptr())
        del t
    # the last entries still answer for tensors that share their storage only; a NEW tensor never hits by accident
    for i in range(10):
        u = torch.full((4, 4), 123.0, dtype=torch.float64)
        hit = hostcache.lookup(u)
        assert hit is None, "a fresh tensor must not match a cached entry (cached tensors are kept alive)"
    assert len(hostcache._entries) <= hostcache._CAP
    big = torch.zeros(100)
    hostcache.remember(big, big.numpy())  # only small tensors are cached
    assert hostcache.lookup(big) is None
