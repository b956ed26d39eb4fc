#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for d in 0; do
rm -rf $R/gpurun_out/prof5 && mkdir -p $R/gpurun_out/prof5
PIN_GQ_DBG=$d rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof5 -o t -- python $R/scratch/time5.py > $R/gpurun_out/prof5/log.txt 2>&1
python - <<PY
import csv, collections
rows=list(csv.DictReader(open("$R/gpurun_out/prof5/t_kernel_trace.csv")))
d=collections.defaultdict(list)
for r in rows:
    if "gn_iteration" in r["Kernel_Name"] or "gn_accumulate" in r["Kernel_Name"]: d[r["Kernel_Name"][:70]].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in d.items():
    n=len(v)//2
    print("DBG=$d", k, [round(sum(v[i*n+5:(i+1)*n])/(n-5)/1e3,1) for i in range(2)])
PY
done
