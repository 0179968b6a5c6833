#!/bin/bash
B="--no-strict --no-host-path --no-cpu-baseline"
NBE_WINO=0 timeout -k 10 250 python bench.py $B > gpurun_out/w6_off.json 2> gpurun_out/w6_off.err
timeout -k 10 250 python bench.py $B > gpurun_out/w6_on.json 2> gpurun_out/w6_on.err
for v in ASMLD; do
  NBE_LIB=$PWD/jax_nbody_emulator_with_dj_amd/libnbe_$v.so timeout -k 10 250 python bench.py $B > gpurun_out/w6_$v.json 2> gpurun_out/w6_$v.err
done
timeout -k 10 250 python bench.py $B > gpurun_out/w6_on2.json 2> gpurun_out/w6_on2.err
python - <<'PY'
import json
for v in ("off", "on", "ASMLD", "on2"):
    try:
        d = json.load(open("gpurun_out/w6_%s.json" % v))
        print(v, round(d["ms_per_step"], 1), d["finite"], [(k["kernel"][:8], round(k["ms"] / 2, 1), k["launches"] // 2, k["tflops"]) for k in d["kernels"][:2]])
    except Exception as e:
        print(v, "failed", e)
PY
