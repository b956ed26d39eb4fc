#!/bin/bash
# Round-2 profiles (run on the GPU box through gpurun; results under gpurun_out/prof_r02, summaries copied to profiles/ by
# scripts/profile_r02_summarize.py):
#   1. tracker kernels ALONE at the bench shape (98 756 queries, C3 map): kernel-trace stats -> AverageNs is the number
#   2. the same command under the PMC passes (separate runs, kernel-trace only, never mixed with other trace domains)
#   3. the default bench command: kernel-trace stats + trace gaps
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r02; rm -rf $O; mkdir -p $O
CMD="python $R/scripts/gn_knn_microbench.py 16 98756 50"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tracker -o t -- $CMD > $O/tracker.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/$C -o p -- $CMD > $O/$C.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY --output-format csv -d $O/SQ1 -o p -- $CMD > $O/SQ1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d $O/SQ2 -o p -- $CMD > $O/SQ2.log 2>&1
# 2b. one training step alone (pin_train_step: stage + fused tile kernel + streamed weight gradient + finalize), per batch size
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train16k -o t -- python $R/scripts/train_microbench.py 16 16384 40 > $O/train16k.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train1m -o t -- python $R/scripts/train_microbench.py 16 1048576 10 > $O/train1m.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/T$C -o p -- python $R/scripts/train_microbench.py 16 1048576 6 > $O/T$C.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --c4-iters 0 --skip-downsampled > $O/bench.log 2>&1
python $R/scripts/trace_gaps.py $O/bench/bench_kernel_trace.csv > $O/trace_gaps.txt 2>&1
python $R/scripts/profile_r02_summarize.py $O
