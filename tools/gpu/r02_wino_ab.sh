#!/bin/bash
# same-device A/B of the Winograd-z kernel (NBE_WINO=0 / 1) after the gauged layer tests
B="--no-strict --no-host-path --no-cpu-baseline"
T=${1:-w7}
timeout -k 10 200 python -m pytest tests/test_gpu_layers.py -x -q -m gpu -k "gauged or winograd" > gpurun_out/${T}_layers.log 2>&1; tail -2 gpurun_out/${T}_layers.log
NBE_WINO=0 timeout -k 10 250 python bench.py $B > gpurun_out/${T}_off.json 2> gpurun_out/${T}_off.err
timeout -k 10 250 python bench.py $B > gpurun_out/${T}_on.json 2> gpurun_out/${T}_on.err
python - $T <<'PY'
import json, sys
T = sys.argv[1]
for v in ("off", "on"):
    try:
        d = json.load(open("gpurun_out/%s_%s.json" % (T, v)))
        print(v, round(d["ms_per_step"], 1), d["finite"], [(k["kernel"][:8], round(k["ms"] / 2, 1), k["launches"] // 2, k["tflops"]) for k in d["kernels"][:2]])
    except Exception as e:
        print(v, "failed", e)
PY
