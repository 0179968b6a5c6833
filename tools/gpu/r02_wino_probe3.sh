#!/bin/bash
B="--no-strict --no-host-path --no-cpu-baseline"
timeout -k 10 200 python -m pytest tests/test_gpu_layers.py -x -q -m gpu -k "gauged or winograd" > gpurun_out/w5_layers.log 2>&1; tail -2 gpurun_out/w5_layers.log
timeout -k 10 250 python bench.py $B > gpurun_out/w5_on.json 2> gpurun_out/w5_on.err
NBE_WINO=0 timeout -k 10 250 python bench.py $B > gpurun_out/w5_off.json 2> gpurun_out/w5_off.err
for v in NOBR NOXF; do
  NBE_LIB=$PWD/jax_nbody_emulator_with_dj_amd/libnbe_$v.so timeout -k 10 250 python bench.py $B > gpurun_out/w5_$v.json 2> gpurun_out/w5_$v.err
done
python - <<'PY'
import json
for v in ("off", "on", "NOBR", "NOXF"):
    try:
        d = json.load(open("gpurun_out/w5_%s.json" % v))
        print(v, round(d["ms_per_step"], 1), d["finite"], [(k["kernel"][:8], round(k["ms"] / 2, 1), k["launches"] // 2, k["tflops"]) for k in d["kernels"][:2]])
    except Exception as e:
        print(v, "failed", e)
PY
