"""Phase times of the training tile kernel's first tile per wave (debug build: scripts/build_variant.sh stamps -DPIN_TF_STAMPS=<wave>):
runs the C3 bench for a few frames, then reads the stamps of the LAST launch (a C3-shape mapping iteration)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = ["bench.py", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-parity", "--c4-iters", "0", "--skip-downsampled",
            "--events", "none", "--moving-steps", "0", "--mesher-queries", "0"]
import bench
bench.main()
from pin_slam_amd import _lib
dll = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * (1024 * 16))()
assert dll.pin_debug_tf_stamps(buf, 1024 * 16) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 16).astype(np.int64)[:256, :8]
ok = (a[:, 7] > a[:, 0]) & (a[:, 0] > 0)
a = a[ok]
names = ["start -> loads requested", "-> image in LDS (barrier)", "-> gather arithmetic done", "-> forward layers", "-> head + loss",
         "-> backward sweep", "-> scatter issued"]
d = np.diff(a, axis=1) * 10.0 / 1e3  # 100 MHz ticks -> us
print(f"blocks with a stamped tile: {len(a)}; whole first tile {np.mean((a[:, 7] - a[:, 0]) * 0.01):.2f} us (median {np.median((a[:, 7] - a[:, 0]) * 0.01):.2f})")
for i, n in enumerate(names):
    print(f"  {n:32s} mean {d[:, i].mean():6.2f} us   median {np.median(d[:, i]):6.2f}   p90 {np.percentile(d[:, i], 90):6.2f}")
span = (a[:, 7].max() - a[:, 0].min()) * 0.01
print(f"first start -> last scatter over the blocks: {span:.2f} us; starts spread over {(a[:, 0].max() - a[:, 0].min()) * 0.01:.2f} us")
