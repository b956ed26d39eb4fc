"""Host time of C-ABI calls by their launch count (perf_counter around the ctypes call, GPU idle before each: the queue is
never full): where the host-bound stages' time goes."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pin_slam_amd import _lib, ops, synth
from pin_slam_amd._lib import check
L = _lib.lib()
m = synth.build_map(layers=3, radius=40.0, raw_per_layer=400_000)
scan = torch.from_numpy(synth.make_scan(m, n=100_000, seed=3)).cuda().contiguous()
n = scan.shape[0]
ws = torch.empty((int(L.pin_maint_workspace_bytes(n)),), dtype=torch.uint8, device="cuda")
sel = torch.empty((n,), dtype=torch.int32, device="cuda")
cnt = torch.zeros((1,), dtype=torch.int32, device="cuda")
out = torch.empty_like(scan)
st = ops._stream()
T = np.eye(4)[:3, :4].copy()

def t(fn, reps=200):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        a = time.perf_counter(); fn(); ts.append(time.perf_counter() - a)
    ts.sort()
    return 1e6 * ts[len(ts) // 2]

print("transform_points (1 launch)          %.1f us" % t(lambda: check(L.pin_transform_points(scan.data_ptr(), 3, n, T.ctypes.data, out.data_ptr(), st), "x")))
print("voxel_downsample_fast (5 + sort)     %.1f us" % t(lambda: check(L.pin_voxel_downsample_fast(scan.data_ptr(), n, 0.08, sel.data_ptr(), cnt.data_ptr(), ws.data_ptr(), ws.numel(), st), "x")))
print("spatial_sort (2 + sort)              %.1f us" % t(lambda: check(L.pin_spatial_sort(scan.data_ptr(), n, 0.1, out.data_ptr(), None, ws.data_ptr(), ws.numel(), st), "x")))
print("torch.randn(100000, 1)               %.1f us" % t(lambda: torch.randn(100000, 1, device="cuda")))
print("torch.empty((n,3))                   %.1f us" % t(lambda: torch.empty((n, 3), device="cuda")))
print("x.zero_()                            %.1f us" % t(lambda: cnt.zero_()))
print("ops._stream()                        %.1f us" % t(lambda: ops._stream()))
s2 = torch.cuda.Stream()
def ctx():
    with torch.cuda.stream(s2):
        pass
print("with torch.cuda.stream(side): pass   %.1f us" % t(ctx))
print("side.wait_stream(main)               %.1f us" % t(lambda: s2.wait_stream(torch.cuda.current_stream())))
ev = torch.cuda.Event()
print("event.record + wait_event            %.1f us" % t(lambda: (ev.record(), s2.wait_event(ev))))
print("cnt.item() (sync read-back)          %.1f us" % t(lambda: cnt.item()))
