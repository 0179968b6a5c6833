#!/bin/bash
# centre-tap pairing across groups in conv_h3g_kernel: parity, then A/B against the previous build (libnbe_prev.so) on one device
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_api.py -x -q -m gpu > gpurun_out/r02_pair_tests.log 2>&1
rc=$?; tail -3 gpurun_out/r02_pair_tests.log
[ $rc -ne 0 ] && exit $rc
for v in new prev new prev; do
  if [ $v = prev ]; then export NBE_LIB=$PWD/jax_nbody_emulator_with_dj_amd/libnbe_prev.so; else unset NBE_LIB; fi
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-strict --no-host-path > gpurun_out/r02_ab_pair_$v.json 2> gpurun_out/r02_ab_pair_$v.err || exit 1
  python - $v <<PY
import json, sys
d=json.loads([l for l in open("gpurun_out/r02_ab_pair_%s.json" % sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[1], round(d["value"]/1e6,2), "Mvox/s", round(d["ms_per_step"],1), "ms", d["roofline"]["kernel"], round(d["roofline"]["frac"],4), round(d["roofline"]["avg_launch_ms"],3))
PY
done
