#!/bin/bash
# round 2 GPU call: the new range / config tests, the host pipeline test, the whole GPU suite, a bench line
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_range.py -x -q -s -m gpu > gpurun_out/r02_range.log 2>&1 && \
python -m pytest "tests/test_gpu_api.py::test_host_array_pipeline_equals_resident" -x -q -s -m gpu > gpurun_out/r02_pipe.log 2>&1 && \
python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_range.py > gpurun_out/r02_gpu_all.log 2>&1 && \
python bench.py --steps 3 --warmup 1 > gpurun_out/r02_bench_base.json 2> gpurun_out/r02_bench_base.err
rc=$?
tail -25 gpurun_out/r02_range.log; tail -5 gpurun_out/r02_pipe.log; tail -3 gpurun_out/r02_gpu_all.log; cat gpurun_out/r02_bench_base.json | cut -c1-1500
exit $rc
