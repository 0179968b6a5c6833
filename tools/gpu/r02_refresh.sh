#!/bin/bash
# refresh the schedule fuzz and the per-rank brick timings on the final kernels
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python tools/fuzz_schedules.py > gpurun_out/r02_fuzz_schedules.txt 2>&1; rc=$?; tail -4 gpurun_out/r02_fuzz_schedules.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python tools/time_brick.py > gpurun_out/r02_brick_times_one_card.txt 2>&1; rc=$?; cat gpurun_out/r02_brick_times_one_card.txt
exit $rc
