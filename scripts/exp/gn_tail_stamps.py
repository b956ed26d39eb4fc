"""Where the time of the fused end of a registration iteration goes (library built with PIN_EXTRA_CFLAGS=-DPIN_GN_STAMPS): the
last block of the tile kernel leaves 100 MHz time stamps in the spare slots of the loop state -- tail entered (the block's atomics
are on their way), release fence done, ticket back, solve entered (acquire fence done), every load back, solve done.  bench.py's
frames with PIN_GN_ITERATE=0; the stamps of every registration's LAST iteration, medians in microseconds from the first stamp."""
import os
import runpy
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["PIN_GN_ITERATE"] = "0"
from pin_slam_amd import engine  # noqa: E402

rows = []
_track = engine.GNTracker.track


def track(self, *a, **k):
    out = _track(self, *a, **k)
    s = self.state_host.numpy()
    rows.append([s[73], s[77], s[78], s[74], s[75], s[76]])
    return out


engine.GNTracker.track = track
sys.argv = ["bench.py"] + sys.argv[1:]
try:
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
finally:
    r = np.array(rows[5:])
    if len(r):
        d = (r - r[:, :1]) / 100.0
        names = ["tail entered", "release fence done", "ticket back", "solve entered", "loads back", "solve done"]
        print("stamps (us, median over %d registrations):" % len(r), {n: round(float(np.median(d[:, i])), 2) for i, n in enumerate(names)}, file=sys.stderr)
