#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_sharded.py -x -q -m gpu > gpurun_out/r02_sharded.log 2>&1
rc=$?; tail -5 gpurun_out/r02_sharded.log
[ $rc -ne 0 ] && exit $rc
python tools/time_brick.py > gpurun_out/r02_brick_times.txt 2>&1; cat gpurun_out/r02_brick_times.txt
