"""Timeline of ONE frame of bench.py out of a rocprofv3 kernel trace (csv): every launch of the last complete frame with its
start offset, duration and the idle gap in front of it; the registration loop and the training iterations are collapsed to
one line per kernel class.  Frames are cut at extract_xyz_kernel (the first launch of pin_preprocess_frame).
  usage: python scripts/frame_timeline.py <..._kernel_trace.csv> [--full] [--back=K]   (K = 0: the last complete frame, 1: the one before, ...)
With --preprocess-stream side the scan chain of frame f+1 starts beside the mapping of frame f: the cut is still its first launch."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
marks = [i for i, e in enumerate(ev) if "extract_xyz_kernel" in e[2]]
if len(marks) < 2:
    sys.exit("fewer than two frames in the trace")
per = 2 if "--marks-per-frame=1" not in sys.argv else 1  # (pin_preprocess_frame runs twice per frame: source and map clouds)
marks = marks[::per]
back = next((int(x.split("=")[1]) for x in sys.argv if x.startswith("--back=")), 0)
a, b = marks[-2 - back], marks[-1 - back]
seg = ev[a:b]
t0 = seg[0][0]
wall = seg[-1][1] - t0
busy = sum(e[1] - e[0] for e in seg)
print(f"frame {-1 - back} of the trace: {len(seg)} launches, wall {wall / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms, idle {(wall - busy) / 1e6:.3f} ms")


def short(n):
    n = n.replace("void ", "").replace("pin::", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0][:64]


LOOP = ("knn_brick_listed", "gn_accumulate", "gn_solve", "train_fused", "train_dw", "train_finalize", "adam_lazy_prepare")
agg = collections.OrderedDict()
prev_end = t0
for s, e, n in seg:
    k = short(n)
    gap = s - prev_end
    prev_end = max(prev_end, e)
    if any(x in n for x in LOOP) and "--full" not in sys.argv:
        d = agg.setdefault(k, [0, 0, 0, s])
        d[0] += 1; d[1] += e - s; d[2] += max(gap, 0)
        continue
    print(f"  {(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap / 1e3:7.1f}  {k}")
print("loops (launches, total us, idle in front of them us, first start us):")
for k, (c, d, g, s) in agg.items():
    print(f"  {k:64s} x{c:4d}  {d / 1e3:8.1f}  gaps {g / 1e3:7.1f}  from {(s - t0) / 1e3:8.1f}")
