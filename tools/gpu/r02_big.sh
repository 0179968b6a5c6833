#!/bin/bash
# conv_h3g_kernel with the 4 x 4 wave tile, one wave per SIMD (NBE_H3G_BIG=1): parity, then A/B against the default on one device
set -o pipefail
mkdir -p gpurun_out
NBE_H3G_BIG=1 timeout -k 10 600 python -m pytest tests/test_gpu_model.py -x -q -m gpu > gpurun_out/r02_big_tests.log 2>&1
rc=$?; tail -3 gpurun_out/r02_big_tests.log
[ $rc -ne 0 ] && exit $rc
for v in 1 0 1 0; do
  NBE_H3G_BIG=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-strict --no-host-path > gpurun_out/r02_ab_big_$v.json 2> gpurun_out/r02_ab_big_$v.err || exit 1
  python - $v <<PY
import json, sys
d=json.loads([l for l in open("gpurun_out/r02_ab_big_%s.json" % sys.argv[1]) if l.startswith("{")][-1])
print("BIG=%s" % sys.argv[1], round(d["value"]/1e6,2), "Mvox/s", round(d["ms_per_step"],1), "ms", d["roofline"]["kernel"], round(d["roofline"]["frac"],4), round(d["roofline"]["avg_launch_ms"],3))
PY
done
