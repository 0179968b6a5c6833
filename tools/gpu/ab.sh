#!/bin/bash
# Same-device A/B of one environment switch on the default bench workload.
#   tools/gpu/ab.sh TAG VAR OFF_VALUE ON_VALUE [pytest -k expression run first] [extra bench.py arguments ...]
# e.g. tools/gpu/ab.sh r03_wino NBE_WINO 0 1 "gauged or winograd"
# Writes gpurun_out/${TAG}_{off,on}.json (+ .err) and prints one summary line per leg.  Both legs run in ONE gpurun call,
# i.e. on one device: the pool's devices differ by ~5 % on this workload, so only such pairs are comparable.
set -o pipefail
TAG=${1:?tag}; VAR=${2:?variable}; OFF=${3:?off value}; ON=${4:?on value}; KEXPR=${5:-}; shift 5 2>/dev/null || shift $#
B="--no-strict --no-host-path --no-cpu-baseline --no-small-configs $*"
mkdir -p gpurun_out
if [ -n "$KEXPR" ]; then
    timeout -k 10 400 python -m pytest tests/test_gpu_layers.py tests/test_gpu_model.py -x -q -m gpu -k "$KEXPR" > gpurun_out/${TAG}_tests.log 2>&1 || { tail -20 gpurun_out/${TAG}_tests.log; exit 1; }
    tail -2 gpurun_out/${TAG}_tests.log
fi
env $VAR=$OFF timeout -k 10 300 python bench.py $B > gpurun_out/${TAG}_off.json 2> gpurun_out/${TAG}_off.err && \
env $VAR=$ON timeout -k 10 300 python bench.py $B > gpurun_out/${TAG}_on.json 2> gpurun_out/${TAG}_on.err
rc=$?
python - $TAG <<'PY'
import json, sys
T = sys.argv[1]
for v in ("off", "on"):
    try:
        d = json.loads([l for l in open("gpurun_out/%s_%s.json" % (T, v)) if l.startswith("{")][-1])
        print(v, round(d["ms_per_step"], 1), "ms", round(d["value"] / 1e6, 2), "Mvox/s", d["roofline"]["kernel"][:24], round(d["roofline"]["frac"], 4),
              [(k["kernel"][:10], round(k["ms"], 1), k["launches"]) for k in d["kernels"][:4]])
    except Exception as e:
        print(v, "failed", e)
PY
exit $rc
