#!/bin/bash
# the default bench line (roofline + traffic + host_path + strict_f32 + cpu_baseline), and the A/B switches' parity
set -o pipefail
mkdir -p gpurun_out
python bench.py > gpurun_out/r02_bench_final.json 2> gpurun_out/r02_bench_final.err || { tail -5 gpurun_out/r02_bench_final.err; exit 1; }
tail -c 3000 gpurun_out/r02_bench_final.json
NBE_H3G_TALL=0 NBE_STEM=0 timeout -k 10 600 python -m pytest tests/test_gpu_layers.py tests/test_gpu_model.py -x -q -m gpu > gpurun_out/r02_final_switches_off.log 2>&1
rc=$?; tail -3 gpurun_out/r02_final_switches_off.log
exit $rc
