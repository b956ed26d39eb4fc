#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for d in 0; do
rm -rf $R/gpurun_out/prof4 && mkdir -p $R/gpurun_out/prof4
PIN_GQ_DBG=$d rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof4 -o t -- python $R/scratch/time4.py > $R/gpurun_out/prof4/log.txt 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/prof4/t_kernel_trace.csv")))
import collections
d=collections.defaultdict(list)
for r in rows:
    if "gn_accumulate" in r["Kernel_Name"]: d[r["Kernel_Name"][:60]].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in d.items():
    # three decoders, 33 launches each in order
    n=len(v)//3
    print("DBG=$d", k, [round(sum(v[i*n:(i+1)*n])/n/1e3,1) for i in range(3)])
PY
done
