#!/bin/bash
# rocprofv3 kernel-trace stats of the default bench command (shorter run)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof && mkdir -p $R/gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o bench -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof/bench.log 2>&1
python - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/prof/**/*kernel_stats.csv", recursive=True)
rows=list(csv.DictReader(open(f[0])))
for r in rows[:32]: print(r["Name"][:70].ljust(70), r["Calls"].rjust(7), r["TotalDurationNs"].rjust(12), r["AverageNs"].rjust(10), r["Percentage"].rjust(7))
PY
python $R/scripts/trace_gaps.py $R/gpurun_out/prof/bench_kernel_trace.csv
tail -1 $R/gpurun_out/prof/bench.log | cut -c1-300
