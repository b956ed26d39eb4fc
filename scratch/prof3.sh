cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for M in full nodec; do
rm -rf $R/gpurun_out/prof3 && mkdir -p $R/gpurun_out/prof3
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof3 -o t -- python $R/scratch/time3.py $M > $R/gpurun_out/prof3/log.txt 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/prof3/t_kernel_stats.csv")))
print("== $M")
for r in rows[:9]: print(r["Name"][:70].ljust(70), r["Calls"].rjust(6), r["AverageNs"].rjust(12))
PY
done
