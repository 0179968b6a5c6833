#!/bin/bash
# A/B of BASELINE config 1 against the float64 fixture: narrow head kernel on / off, strict f32
mkdir -p gpurun_out
for nv in 1 0; do
  NBE_NARROW=$nv python -m pytest "tests/test_gpu_range.py::test_config1_apply_full_width_matches_oracle" -q -s -m gpu > gpurun_out/r02_c1_narrow$nv.log 2>&1
  echo "NBE_NARROW=$nv rc=$?"; grep -h "config 1\|AssertionError:" gpurun_out/r02_c1_narrow$nv.log | grep -v "^ " | head -8
done
