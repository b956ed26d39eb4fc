cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof2 && mkdir -p $R/gpurun_out/prof2
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof2 -o t -- python $R/scratch/time2.py > $R/gpurun_out/prof2/log.txt 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/prof2/t_kernel_stats.csv")))
for r in rows[:12]: print(r["Name"][:90].ljust(90), r["Calls"].rjust(6), r["AverageNs"].rjust(12))
PY
