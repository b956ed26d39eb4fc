"""Summaries of scripts/profile_r02.sh: per-kernel stats of the tracker-alone run and of the bench command, and the PMC
counters per launch of the two tracker kernels (written next to the raw output; copy into profiles/)."""
import collections, csv, glob, json, os, sys

O = sys.argv[1]


def stats(sub, out, top=40):
    f = glob.glob(os.path.join(O, sub, "**", "*kernel_stats.csv"), recursive=True)
    if not f:
        return
    rows = list(csv.DictReader(open(f[0])))
    with open(os.path.join(O, out), "w") as w:
        wr = csv.writer(w)
        wr.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows[:top]:
            wr.writerow([r["Name"][:110], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])
    for r in rows[:12]:
        print(sub, r["Name"][:70].ljust(70), r["Calls"].rjust(7), r["AverageNs"].rjust(10), r["Percentage"].rjust(7))


stats("tracker", "r02_tracker_kernel_stats.csv")
stats("bench", "r02_bench_kernel_stats.csv")
stats("train16k", "r02_train_16k_kernel_stats.csv", top=8)
stats("train1m", "r02_train_1m_kernel_stats.csv", top=8)
res = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ1", "SQ2", "TFETCH_SIZE", "TWRITE_SIZE"):
    for f in glob.glob(os.path.join(O, c, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            key = ("knn_brick" if "knn_brick" in n else "gn" if "gn_accumulate" in n else "train_fused" if "train_fused" in n
                   else "train_dw_stream" if "train_dw_stream" in n else None)
            if c.startswith("T") and key not in ("train_fused", "train_dw_stream"):
                continue
            if key:
                res[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"command": "scripts/profile_r02.sh: rocprofv3 --kernel-trace --pmc <set> -- python scripts/gn_knn_microbench.py 16 98756 50 "
                  "(separate passes for FETCH_SIZE, WRITE_SIZE and two SQ sets)",
       "workload": "bench C3 map (2.23 M neural points), 98 756 Morton-ordered scan points, Kc = 81, k = 8, decoder 4x64; "
                   "55 launches of each kernel, all of one shape",
       "units": "FETCH_SIZE / WRITE_SIZE in KiB per launch (narrow random reads: counted at face value, see r01_pmc.json "
                "calibration); SQ_* summed over the chip per launch",
       "kernels": {k: {c: round(sum(v) / len(v), 1) for c, v in d.items()} for k, d in res.items()}}
for k in ("gn", "knn_brick", "train_fused", "train_dw_stream"):
    d = out["kernels"].get(k, {})
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        out[f"{k}_hbm_bytes_per_launch"] = int((d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024)
json.dump(out, open(os.path.join(O, "r02_pmc.json"), "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
