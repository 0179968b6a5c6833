#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_range.py -x -q -m gpu > gpurun_out/r02_eighth_tests.log 2>&1
rc=$?; tail -3 gpurun_out/r02_eighth_tests.log
[ $rc -ne 0 ] && exit $rc
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-strict --no-host-path > gpurun_out/r02_bench_l0flat.json 2> gpurun_out/r02_bench_l0flat.err && \
NBE_L0_FLAT=0 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-strict --no-host-path > gpurun_out/r02_bench_l0patch.json 2> gpurun_out/r02_bench_l0patch.err
rc=$?
for f in l0flat l0patch; do python - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/r02_bench_$f.json") if l.startswith("{")][-1])
    print("$f", round(d["value"]/1e6,2), "Mvox/s", round(d["ms_per_step"],1), "ms", d["roofline"]["kernel"], round(d["roofline"]["frac"],4))
    for k in d["kernels"][:7]: print("   ", k)
except Exception as e: print("$f", e)
PY
done
exit $rc
