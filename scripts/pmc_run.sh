#!/bin/bash
# Separate PMC passes (never combined with other trace domains), kernel-trace only.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc; rm -rf $O; mkdir -p $O
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/$C -o p -- python $R/scripts/pmc_workload.py 16 > $O/$C.log 2>&1
done
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/TCC -o p -- python $R/scripts/pmc_workload.py 16 > $O/TCC.log 2>&1
python - <<PY
import csv, glob, json, collections
res = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE", "TCC"):
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            key = "knn" if "knn_query" in n else "gn" if "gn_accumulate" in n else "copy" if ("copy" in n.lower() or "elementwise" in n.lower()) else None
            if key: res[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, d in res.items():
    out[k] = {c: {"n": len(v), "mean": sum(v) / len(v), "max": max(v)} for c, v in d.items()}
json.dump(out, open("$O/pmc_raw.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
