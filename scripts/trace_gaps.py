"""Analyse a rocprofv3 kernel trace of bench.py: per-frame GPU busy time and idle gaps."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
# frames: split at brick_mark (start of reset_local_map in process_frame) -- use vds_init as frame marker (first kernel of preprocess)
marks = [i for i, e in enumerate(ev) if "vds_init" in e[2]]
# 3 voxel downsamples per frame (2 preprocess + 1 update); take every 3rd as frame start, from the end
frames = marks[::3]
print("events", len(ev), "vds_init", len(marks))
for a, b in list(zip(frames[:-1], frames[1:]))[-8:]:
    seg = ev[a:b]
    wall = seg[-1][1] - seg[0][0]
    busy = sum(e[1] - e[0] for e in seg)
    gaps = collections.Counter()
    for x, y in zip(seg[:-1], seg[1:]):
        g = y[0] - x[1]
        if g > 20000:
            gaps[(x[2][:40], y[2][:40])] += g
    print(f"frame kernels={len(seg)} wall={wall/1e6:.3f} ms busy={busy/1e6:.3f} ms idle={(wall-busy)/1e6:.3f} ms")
    for k, v in gaps.most_common(8):
        print(f"    gap {v/1e3:8.1f} us  {k[0]} -> {k[1]}")

# the mapping stage of the last complete frame in detail: from the gather launch of its first group of iterations to the
# optimiser's flush (start offset us, duration us, gap before us)
g = [i for i, e in enumerate(ev) if "gather_batch_drawn" in e[2]]
f = [i for i, e in enumerate(ev) if "adam_lazy_flush" in e[2]]
if g and f:
    b = f[-1]
    a = max(i for i in g if i < b)
    while a - 1 in g:
        a -= 1
    print("one Mapper.mapping call (start offset us, duration us, gap before us):")
    t0 = ev[a][0]
    for i in range(a, b + 1):
        print(f"    {(ev[i][0]-t0)/1e3:8.1f} {(ev[i][1]-ev[i][0])/1e3:7.1f} {(ev[i][0]-ev[i-1][1])/1e3:7.1f}  {ev[i][2][:70]}")

# every launch of the last complete frame outside the GN loop and the training loop (start offset us, duration us, gap before us)
if "--frame" in sys.argv and len(frames) >= 2:
    a, b = frames[-2], frames[-1]
    t0 = ev[a][0]
    skip = ("knn_brick", "gn_accumulate", "gn_solve", "train_fused", "train_dw", "train_finalize", "adam_lazy_prepare")
    print("last frame, launches outside the two hot loops:")
    for i in range(a, b):
        if any(k in ev[i][2] for k in skip):
            continue
        print(f"    {(ev[i][0]-t0)/1e3:8.1f} {(ev[i][1]-ev[i][0])/1e3:7.1f} {(ev[i][0]-ev[i-1][1])/1e3:7.1f}  {ev[i][2][:90]}")
