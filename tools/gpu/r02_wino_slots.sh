#!/bin/bash
B="--no-strict --no-host-path --no-cpu-baseline"
timeout -k 10 200 python -m pytest tests/test_gpu_layers.py tests/test_gpu_model.py -x -q -m gpu -k "gauged or small" 2>&1 | tail -2
for v in new OLDSLOTS new2 OLDSLOTS2; do
  L=$PWD/jax_nbody_emulator_with_dj_amd/libnbe.so; case $v in OLD*) L=$PWD/jax_nbody_emulator_with_dj_amd/libnbe_OLDSLOTS.so;; esac
  NBE_LIB=$L NBE_PROF_LAYERS=1 timeout -k 10 250 python bench.py $B > gpurun_out/w18_$v.json 2> gpurun_out/w18_$v.err
done
python - <<'PY'
import json
for v in ("new", "OLDSLOTS", "new2", "OLDSLOTS2"):
    try:
        d = json.load(open("gpurun_out/w18_%s.json" % v))
        ks = {k["kernel"].split(" ", 1)[-1]: k for k in d["kernels"]}
        print(v, round(d["ms_per_step"], 1), d["finite"], [(n, round(ks[n]["ms"] / 2, 1), ks[n]["tflops"]) for n in ("conv_r00/conv_0", "conv_l01/conv_0", "conv_l01/conv_1", "conv_r00/conv_1")])
    except Exception as e:
        print(v, "failed", e)
PY
