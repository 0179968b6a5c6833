#!/bin/bash
# round 2: host pipeline test, whole GPU suite (without the range file, run before), bench with A/B of the new kernels
set -o pipefail
mkdir -p gpurun_out
python -m pytest "tests/test_gpu_api.py::test_host_array_pipeline_equals_resident" -x -q -s -m gpu > gpurun_out/r02_pipe.log 2>&1 && \
python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_range.py > gpurun_out/r02_gpu_all.log 2>&1 && \
python bench.py --steps 3 --warmup 1 > gpurun_out/r02_bench_fused.json 2> gpurun_out/r02_bench_fused.err && \
NBE_FUSE=0 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-strict --no-host-path > gpurun_out/r02_bench_nofuse.json 2> gpurun_out/r02_bench_nofuse.err && \
NBE_FUSE=0 NBE_NARROW=0 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-strict --no-host-path > gpurun_out/r02_bench_nofuse_nonarrow.json 2> gpurun_out/r02_bench_nofuse_nonarrow.err
rc=$?
tail -5 gpurun_out/r02_pipe.log; tail -3 gpurun_out/r02_gpu_all.log
for f in fused nofuse nofuse_nonarrow; do python - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/r02_bench_$f.json") if l.startswith("{")][-1])
    print("$f", round(d["value"]/1e6,2), "Mvox/s", round(d["ms_per_step"],1), "ms", d["roofline"]["kernel"], round(d["roofline"]["frac"],4), d.get("host_path"))
    for k in d["kernels"][:8]: print("   ", k)
except Exception as e: print("$f", e)
PY
done
exit $rc
