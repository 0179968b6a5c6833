#!/bin/bash
# A/B of the 4 x 2 wave tile of conv_h3g_kernel (NBE_H3G_TALL=1, default) against the 2 x 4 tile, one device
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_layers.py tests/test_gpu_model.py -x -q -m gpu > gpurun_out/r02_tall_tests.log 2>&1
rc=$?; tail -3 gpurun_out/r02_tall_tests.log
[ $rc -ne 0 ] && exit $rc
for v in 1 0 1 0; do
  NBE_H3G_TALL=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-strict --no-host-path > gpurun_out/r02_ab_tall_$v.json 2> gpurun_out/r02_ab_tall_$v.err || exit 1
  python - $v <<PY
import json, sys
d=json.loads([l for l in open("gpurun_out/r02_ab_tall_%s.json" % sys.argv[1]) if l.startswith("{")][-1])
print("TALL=%s" % sys.argv[1], round(d["value"]/1e6,2), "Mvox/s", round(d["ms_per_step"],1), "ms", d["roofline"]["kernel"], round(d["roofline"]["frac"],4), round(d["roofline"]["avg_launch_ms"],3))
PY
done
