"""gpurun_out/pmc_bench/<workload>/ (scripts/pmc_bench.sh) -> gpurun_out/pmc_bench/<round>_pmc_<workload>.json and
<round>_bench_<workload>_kernel_stats.csv (PIN_ROUND, default r06) (copy both into profiles/).

Per kernel CLASS and launch SHAPE (grid size): mean counter values per launch.  The tracker's launches of a kernel are the
shape with the most launches (50 per frame); the training launches of the search kernel have other grids.
HBM bytes per launch = (FETCH_SIZE + WRITE_SIZE) KiB -> bytes; the narrow random reads of these kernels are counted at
face value (calibration in profiles/r01_pmc.json: FETCH_SIZE ~= TCC_MISS x 64 B for them; only wide streaming reads show
up halved on gfx950, /opt/skills/guides/MI355X_MICROARCH.md, HBM section -- `fetch_streaming_x2` gives that bound too)."""
import collections, csv, glob, json, os, shutil, sys
RND = os.environ.get("PIN_ROUND", "r06")

W = sys.argv[1]
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
O = os.path.join(R, "gpurun_out", "pmc_bench", W)
CLOCK_GHZ, N_SIMD = 2.4, 1024

CLASSES = (("sdf_query_quad", "sdf_query_quad_kernel"), ("gn", "gn_accumulate"), ("knn_brick_listed", "knn_brick_listed_kernel"), ("knn_brick", "knn_brick_kernel"), ("train_fused", "train_fused"), ("train_dw_recompute", "train_dw_recompute"), ("train_dw_stream", "train_dw_stream"),
           ("adam_lazy_prepare_rows", "adam_lazy_prepare_rows"), ("mark_rows", "mark_rows"), ("adam_lazy_prepare", "adam_lazy_prepare"),
           ("gn_solve", "gn_solve"))
LARGEST = W == "c4"  # the 2^20-sample mapper: its launches are the largest grids of their kernels, not the most frequent


def cls(name):
    for k, pat in CLASSES:
        if pat in name:
            return k
    return None


res = collections.defaultdict(lambda: collections.defaultdict(list))   # (class, grid) -> counter -> values
names = {}
for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ1", "SQ2"):
    for f in glob.glob(os.path.join(O, c, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = cls(r["Kernel_Name"])
            if k is None:
                continue
            key = (k, r.get("Grid_Size", r.get("Grid_Size_X", "?")))
            res[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            names[key] = r["Kernel_Name"][:120]
# durations per (class, grid) from the stats run's trace
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(O, "stats", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = cls(r["Kernel_Name"])
        if k:
            g = str(int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)) if "Grid_Size_X" in r else r.get("Grid_Size", "?")
            dur[(k, g)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = {"command": ("scripts/pmc_bench.sh mesher: rocprofv3 --kernel-trace --pmc <set> -- python bench.py --workload c3 --steps 1 --warmup 0 "
                   "--no-cpu-baseline --no-parity --c4-iters 0 --skip-downsampled --events none --moving-steps 0 --mesher-queries 10000000; the "
                   "most frequent launch shape of knn_brick / sdf_query_quad = one 524 288-query batch of Mesher.query_points" if W == "mesher" else
                   f"scripts/pmc_bench.sh {W}: rocprofv3 --kernel-trace --pmc <set> -- python bench.py --workload {W} --steps 3 --warmup 1 "
                   "--no-cpu-baseline --no-parity --c4-iters 0 --skip-downsampled --events none" if not LARGEST else
                   "scripts/pmc_bench.sh c4: rocprofv3 --kernel-trace --pmc <set> -- python bench.py --workload c3 --steps 1 --warmup 0 "
                   "--no-cpu-baseline --no-parity --c4-iters 6 --dp-emulate= --skip-downsampled --events none; per kernel class the LARGEST "
                   "launch shape = the 2^20-sample Mapper.mapping iterations (the search kernel's launch covers a group of iterations)") +
                  " (separate passes: FETCH_SIZE, WRITE_SIZE, two SQ sets; durations from a --kernel-trace --stats pass of the same command)",
       "units": "FETCH_SIZE / WRITE_SIZE in KiB per launch; SQ_* summed over the chip per launch; duration_us = mean kernel time of "
                "that launch shape in the stats pass",
       "kernels": {}}
by_class = collections.defaultdict(list)
for (k, g), d in res.items():
    by_class[k].append((len(next(iter(d.values()))), g))
for k, shapes in by_class.items():
    n, g = max(shapes)  # the shape with the most launches = the hot loop's
    if LARGEST:
        n, g = max(shapes, key=lambda t: int(t[1]) if str(t[1]).isdigit() else 0)
    d = res[(k, g)]
    top = None
    if LARGEST and k == "train_fused":
        # persistent blocks: one grid for every batch size.  The 2^20-sample launches are the heaviest ones of the run,
        # as many as the weight-gradient kernel has launches of its largest grid
        big = "train_dw_recompute" if any(kk == "train_dw_recompute" for (kk, gg) in res) else "train_dw_stream"  # (large batches)
        dw = [(int(gg), len(next(iter(res[(big, gg)].values())))) for (kk, gg) in res if kk == big and str(gg).isdigit()]
        top = max(dw)[1] if dw else None
    pick = (lambda v: sorted(v, reverse=True)[:top]) if top else (lambda v: v)
    e = {c: round(sum(pick(v)) / len(pick(v)), 1) for c, v in d.items()}
    if top:
        n = top
    e["launch_shape_grid"], e["launches_counted"], e["kernel"] = g, n, names[(k, g)]
    dd = dur.get((k, g)) or [x for (kk, gg), v in dur.items() if kk == k for x in v]
    if dd and top:
        dd = sorted(dd, reverse=True)[:top]
    if dd:
        e["duration_us"] = round(sum(dd) / len(dd) / 1e3, 2)
        cyc = e["duration_us"] * 1e-6 * CLOCK_GHZ * 1e9 * N_SIMD
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e:
            e["mfma_util"] = round(e["SQ_VALU_MFMA_BUSY_CYCLES"] / cyc, 4)       # matrix-pipe busy cycles / SIMD-cycles
        if "SQ_ACTIVE_INST_VALU" in e:
            e["valu_active"] = round(4.0 * e["SQ_ACTIVE_INST_VALU"] / cyc, 4)    # (quad-cycles) vector ALU issuing / SIMD-cycles
        if "SQ_WAIT_ANY" in e and "SQ_WAVES" in e:
            e["wave_wait_share"] = round(4.0 * e["SQ_WAIT_ANY"] / (cyc * max(1.0, e["SQ_WAVES"] / N_SIMD)), 4)
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        e["hbm_bytes_per_launch"] = int((e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024)
        e["hbm_bytes_per_launch_if_streaming_x2"] = int((2 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024)
    out["kernels"][k] = e
dst = os.path.join(R, "gpurun_out", "pmc_bench")
json.dump(out, open(os.path.join(dst, f"{RND}_pmc_{W}.json"), "w"), indent=1)
f = glob.glob(os.path.join(O, "stats", "**", "*kernel_stats.csv"), recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    with open(os.path.join(dst, f"{RND}_bench_{W}_kernel_stats.csv"), "w") as w:
        wr = csv.writer(w)
        wr.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows[:40]:
            wr.writerow([r["Name"][:110], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])
print(W, {k: {c: v for c, v in e.items() if c in ("duration_us", "mfma_util", "valu_active", "hbm_bytes_per_launch", "launches_counted")}
          for k, e in out["kernels"].items()})
