python -m pytest tests/test_gpu_color.py tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_process.py tests/test_gpu_dp.py -x -q 2>&1 | tail -5
for W in c5 c3; do python bench.py --workload $W --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$W', d['value'], d['stage_ms_per_frame'], d['roofline']['kernel'][:30], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline_knn']['avg_launch_ms'])"; done
python scripts/gn_knn_microbench.py 16 2>&1 | grep -v amdgpu
