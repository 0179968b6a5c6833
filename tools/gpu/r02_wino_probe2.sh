#!/bin/bash
# timing probes of conv_h3w_kernel (results invalid): NOST no plane transform at all, NOLD no global loads of the raw planes,
# NOBR unconditional prefetch (no branches around the DMA / load slots)
B="--no-strict --no-host-path --no-cpu-baseline"
for v in NOST NOLD NOBR; do
  NBE_LIB=$PWD/jax_nbody_emulator_with_dj_amd/libnbe_$v.so timeout -k 10 250 python bench.py $B > gpurun_out/w4_$v.json 2> gpurun_out/w4_$v.err
done
timeout -k 10 250 python bench.py $B > gpurun_out/w4_on.json 2> gpurun_out/w4_on.err
python - <<'PY'
import json
for v in ("on", "NOST", "NOLD", "NOBR"):
    try:
        d = json.load(open("gpurun_out/w4_%s.json" % v))
        print(v, round(d["ms_per_step"], 1), d["finite"], [(k["kernel"][:8], round(k["ms"] / 2, 1), k["launches"] // 2, k["tflops"]) for k in d["kernels"][:2]])
    except Exception as e:
        print(v, "failed", e)
PY
