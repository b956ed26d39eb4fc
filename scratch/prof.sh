#!/bin/bash
# rocprofv3 kernel-trace stats of the default bench command (shorter run)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof && mkdir -p $R/gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof/bench.log 2>&1
ls -R $R/gpurun_out/prof | head -20
python - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/prof/**/*kernel_stats.csv", recursive=True)
print(f)
rows=list(csv.DictReader(open(f[0])))
for r in rows[:25]: print(r["Name"][:70].ljust(70), r["Calls"].rjust(7), r["TotalDurationNs"].rjust(12), r["AverageNs"].rjust(10), r["Percentage"].rjust(7))
PY
tail -2 $R/gpurun_out/prof/bench.log | cut -c1-400
