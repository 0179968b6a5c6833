#!/bin/bash
# the A/B switches' paths must stay correct: model + layer parity with the 2 x 4 wave tile and the general first-layer kernel
set -o pipefail
mkdir -p gpurun_out
NBE_H3G_TALL=0 NBE_STEM=0 timeout -k 10 600 python -m pytest tests/test_gpu_layers.py tests/test_gpu_model.py -x -q -m gpu > gpurun_out/r02_sw_off.log 2>&1; rc=$?; echo "TALL=0 STEM=0 rc=$rc"; tail -3 gpurun_out/r02_sw_off.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python -m pytest tests/test_gpu_layers.py tests/test_gpu_model.py -x -q -m gpu > gpurun_out/r02_sw_default.log 2>&1; rc=$?; echo "default rc=$rc"; tail -3 gpurun_out/r02_sw_default.log
exit $rc
