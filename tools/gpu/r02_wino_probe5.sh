#!/bin/bash
B="--no-strict --no-host-path --no-cpu-baseline"
NBE_PROF_LAYERS=1 timeout -k 10 250 python bench.py $B > gpurun_out/w16_on.json 2> gpurun_out/w16_on.err
NBE_PROF_LAYERS=1 NBE_LIB=$PWD/jax_nbody_emulator_with_dj_amd/libnbe_NOEPI.so timeout -k 10 250 python bench.py $B > gpurun_out/w16_NOEPI.json 2> gpurun_out/w16_NOEPI.err
python - <<'PY'
import json
for v in ("on", "NOEPI"):
    try:
        d = json.load(open("gpurun_out/w16_%s.json" % v))
        ks = {k["kernel"].split(" ", 1)[-1]: k for k in d["kernels"]}
        print(v, round(d["ms_per_step"], 1), [(n, round(ks[n]["ms"] / 2, 1), ks[n]["tflops"]) for n in ("conv_r00/conv_0", "conv_l01/conv_0", "conv_l01/conv_1", "conv_r00/conv_1")])
    except Exception as e:
        print(v, "failed", e)
PY
