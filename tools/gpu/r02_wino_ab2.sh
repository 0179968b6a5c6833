#!/bin/bash
# model-level parity (fused skips on the Winograd-z kernel), then the same-device A/B
B="--no-strict --no-host-path --no-cpu-baseline"
T=${1:-w12}
timeout -k 10 400 python -m pytest tests/test_gpu_model.py tests/test_gpu_layers.py -x -q -m gpu > gpurun_out/${T}_tests.log 2>&1; tail -3 gpurun_out/${T}_tests.log
NBE_WINO=0 timeout -k 10 250 python bench.py $B > gpurun_out/${T}_off.json 2> gpurun_out/${T}_off.err
timeout -k 10 250 python bench.py $B > gpurun_out/${T}_on.json 2> gpurun_out/${T}_on.err
python - $T <<'PY'
import json, sys
T = sys.argv[1]
for v in ("off", "on"):
    try:
        d = json.load(open("gpurun_out/%s_%s.json" % (T, v)))
        print(v, round(d["ms_per_step"], 1), d["finite"], [(k["kernel"][:8], round(k["ms"] / 2, 1), k["launches"] // 2, k["tflops"]) for k in d["kernels"][:3]])
    except Exception as e:
        print(v, "failed", e)
PY
