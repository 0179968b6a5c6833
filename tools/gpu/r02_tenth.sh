#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_layers.py tests/test_gpu_model.py -x -q -m gpu > gpurun_out/r02_tenth_tests.log 2>&1
rc=$?; tail -3 gpurun_out/r02_tenth_tests.log
[ $rc -ne 0 ] && exit $rc
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-strict --no-host-path > gpurun_out/r02_bench_ring4.json 2> gpurun_out/r02_bench_ring4.err
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r02_bench_ring4.json") if l.startswith("{")][-1])
print(round(d["value"]/1e6,2), "Mvox/s", round(d["ms_per_step"],1), "ms", d["roofline"]["kernel"], round(d["roofline"]["frac"],4))
for k in d["kernels"][:8]: print("   ", k)
PY
