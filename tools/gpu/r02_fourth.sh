#!/bin/bash
# round 2: MFMA shape rates, up-sample kernel: layer + model tests, bench A/B (NBE_UP8)
set -o pipefail
mkdir -p gpurun_out
./tools/micro/mfma_rate > gpurun_out/r02_mfma_rate.txt 2>&1; cat gpurun_out/r02_mfma_rate.txt
python -m pytest tests/test_gpu_layers.py tests/test_gpu_model.py tests/test_gpu_range.py -x -q -m gpu > gpurun_out/r02_up_tests.log 2>&1
rc=$?; tail -3 gpurun_out/r02_up_tests.log
[ $rc -ne 0 ] && exit $rc
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-strict --no-host-path > gpurun_out/r02_bench_up8.json 2> gpurun_out/r02_bench_up8.err && \
NBE_UP8=0 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-strict --no-host-path > gpurun_out/r02_bench_noup8.json 2> gpurun_out/r02_bench_noup8.err
rc=$?
for f in up8 noup8; do python - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/r02_bench_$f.json") if l.startswith("{")][-1])
    print("$f", round(d["value"]/1e6,2), "Mvox/s", round(d["ms_per_step"],1), "ms", d["roofline"]["kernel"], round(d["roofline"]["frac"],4))
    for k in d["kernels"][:9]: print("   ", k)
except Exception as e: print("$f", e)
PY
done
exit $rc
