#!/bin/bash
# rocprofv3 kernel-trace stats of one command: scripts/prof.sh <out-dir-under-gpurun_out> <rows> <command ...>
# (counters are collected by scripts/pmc_run.sh in runs of their own -- never mixed with other trace domains)
tag=$1; rows=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$tag; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d $O -o p -- "$@" > $O/cmd.log 2>&1 )
python - <<PY
import csv,glob
f=glob.glob("$O/**/*kernel_stats.csv", recursive=True)
rows=list(csv.DictReader(open(f[0])))
tot=sum(int(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.3f ms" % (tot/1e6))
for r in rows[:$rows]: print(r["Name"][:90].ljust(90), r["Calls"].rjust(7), r["TotalDurationNs"].rjust(12), r["AverageNs"].rjust(10), r["Percentage"].rjust(7))
import shutil; shutil.copy(f[0], "$O/../" + "$tag".replace("/","_") + "_kernel_stats.csv")
PY
tail -1 $O/cmd.log | cut -c1-400
