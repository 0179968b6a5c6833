#!/bin/bash
# the default bench line (roofline + traffic + host_path + strict_f32 + cpu_baseline), then the whole GPU suite
set -o pipefail
mkdir -p gpurun_out
python bench.py > gpurun_out/r02_bench_final.json 2> gpurun_out/r02_bench_final.err || { tail -5 gpurun_out/r02_bench_final.err; exit 1; }
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r02_bench_final.json") if l.startswith("{")][-1])
print(round(d["value"]/1e6,2), "Mvox/s", round(d["ms_per_step"],1), "ms", d["roofline"], d["host_path"], d["strict_f32"]["value"], d["cpu_baseline"])
PY
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r02_final_gpu_suite.log 2>&1
rc=$?; tail -4 gpurun_out/r02_final_gpu_suite.log
exit $rc
