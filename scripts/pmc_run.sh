#!/bin/bash
# Separate PMC passes (never combined with other trace domains), kernel-trace only.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc; rm -rf $O; mkdir -p $O
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/$C -o p -- python $R/scripts/pmc_workload.py 16 > $O/$C.log 2>&1
done
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/TCC -o p -- python $R/scripts/pmc_workload.py 16 > $O/TCC.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY --output-format csv -d $O/SQ1 -o p -- python $R/scripts/pmc_workload.py 16 > $O/SQ1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $O/SQ2 -o p -- python $R/scripts/pmc_workload.py 16 > $O/SQ2.log 2>&1
python - <<PY
import csv, glob, json, collections
res = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE", "TCC", "SQ1", "SQ2"):
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            key = "knn_brick" if "knn_brick" in n else "knn" if "knn_query" in n else "gn" if "gn_accumulate" in n else "copy" if ("copy" in n.lower() or "elementwise" in n.lower()) else None
            if key: res[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, d in res.items():
    out[k] = {c: {"n": len(v), "mean": sum(v) / len(v), "max": max(v)} for c, v in d.items()}
json.dump(out, open("$O/pmc_raw.json", "w"), indent=1)
for k in ('knn_brick','knn','gn'):
    print(k, {c: round(v['mean']) for c, v in out.get(k, {}).items()})
PY
