#!/bin/bash
# PMC passes of THE BENCH COMMAND, per workload (run on the GPU box through gpurun): separate rocprofv3 runs for
# FETCH_SIZE, WRITE_SIZE and two SQ sets, kernel-trace only (never combined with other trace domains), plus one
# kernel-trace --stats run for the durations of the same command.  scripts/pmc_bench_summarize.py turns the output
# into profiles/r04_pmc_<workload>.json (read by bench.py for roofline.traffic / mfma_util / valu_active) and
# profiles/r04_bench_<workload>_kernel_stats.csv.
#   usage: scripts/pmc_bench.sh c3 [c2 kitti c5 c4 mesher]   (c4 = the 2^20-sample Mapper.mapping of the C3 map: the summary then takes the
#   LARGEST launch shape of every kernel class instead of the most frequent one)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for W in "$@"; do
  O=$R/gpurun_out/pmc_bench/$W; rm -rf $O; mkdir -p $O
  CMD="python $R/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-parity --c4-iters 0 --skip-downsampled --events none --moving-steps 0 --mesher-queries 0 --semantic-leg 0"
  # mesher = Mesher.query_points over 1e7 grid queries on the C3 map (the forward-only tile decoder + the search through the call's brick cache)
  if [ "$W" = "mesher" ]; then CMD="python $R/bench.py --workload c3 --steps 1 --warmup 0 --no-cpu-baseline --no-parity --c4-iters 0 --skip-downsampled --events none --moving-steps 0 --mesher-queries 10000000 --semantic-leg 0"; fi
  if [ "$W" = "c4" ]; then CMD="python $R/bench.py --workload c3 --steps 1 --warmup 0 --no-cpu-baseline --no-parity --c4-iters 6 --dp-emulate= --skip-downsampled --events none --moving-steps 0 --mesher-queries 0 --semantic-leg 0"; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- $CMD > $O/stats.log 2>&1
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/$C -o p -- $CMD > $O/$C.log 2>&1
  done
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY --output-format csv -d $O/SQ1 -o p -- $CMD > $O/SQ1.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d $O/SQ2 -o p -- $CMD > $O/SQ2.log 2>&1
  python $R/scripts/pmc_bench_summarize.py $W
  # keep the merge small: the raw traces are not needed once summarised
  find $O -name '*kernel_trace.csv' -delete; find $O -name '*counter_collection.csv' -delete; find $O -name '*.db' -delete
done
