#!/bin/bash
# stem_h3_kernel: layer + model parity, then A/B (NBE_STEM=1 default / 0) on one device
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_layers.py tests/test_gpu_model.py -x -q -m gpu > gpurun_out/r02_stem_tests.log 2>&1
rc=$?; tail -5 gpurun_out/r02_stem_tests.log
[ $rc -ne 0 ] && exit $rc
for v in 1 0; do
  NBE_STEM=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-strict --no-host-path > gpurun_out/r02_ab_stem_$v.json 2> gpurun_out/r02_ab_stem_$v.err || exit 1
  python - $v <<PY
import json, sys
d=json.loads([l for l in open("gpurun_out/r02_ab_stem_%s.json" % sys.argv[1]) if l.startswith("{")][-1])
print("STEM=%s" % sys.argv[1], round(d["value"]/1e6,2), "Mvox/s", round(d["ms_per_step"],1), "ms", d["roofline"]["kernel"], round(d["roofline"]["frac"],4))
for k in d["kernels"][:6]: print("   ", k)
PY
done
