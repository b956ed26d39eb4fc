#!/bin/bash
exec < /dev/null
mkdir -p gpurun_out/ab
timeout 200 python bench.py --steps 60 --warmup 20 --no-cpu-baseline --c4-iters 0 --skip-downsampled > gpurun_out/ab/c3.json 2> gpurun_out/ab/c3.err
python - <<PY
import json
d=json.loads(open("gpurun_out/ab/c3.json").read().strip().splitlines()[-1]); print(d["value"], d["stage_ms_per_frame"], d["host_enqueue_ms_per_frame"])
PY
