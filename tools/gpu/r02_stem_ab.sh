#!/bin/bash
# same-device A/B of the stem kernel's per-channel vectors (memory loads inside the epilogue vs LDS-resident)
B="--no-strict --no-host-path --no-cpu-baseline"
for v in new old new2 old2; do
  L=$PWD/jax_nbody_emulator_with_dj_amd/libnbe.so; case $v in old*) L=$PWD/jax_nbody_emulator_with_dj_amd/libnbe_OLDSTEM.so;; esac
  NBE_LIB=$L NBE_PROF_LAYERS=1 timeout -k 10 250 python bench.py $B > gpurun_out/w20_$v.json 2> gpurun_out/w20_$v.err
done
python - <<'PY'
import json
for v in ("new", "old", "new2", "old2"):
    d = json.load(open("gpurun_out/w20_%s.json" % v))
    ks = {k["kernel"].split(" ", 1)[-1]: k for k in d["kernels"]}
    print(v, round(d["ms_per_step"], 1), [(n, round(ks[n]["ms"] / 2, 2)) for n in ("conv_l00/conv_0", "conv_l01/conv_0")])
PY
