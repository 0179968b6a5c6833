#!/bin/bash
# persistent workgroups of conv_h3w_kernel: parity, then the bench for a few grid sizes
B="--no-strict --no-host-path --no-cpu-baseline"
NBE_WINO_PERSIST=256 timeout -k 10 200 python -m pytest tests/test_gpu_layers.py tests/test_gpu_model.py -x -q -m gpu -k "gauged or small" 2>&1 | tail -2
for n in 0 256 512 1024; do
  NBE_WINO_PERSIST=$n NBE_PROF_LAYERS=1 timeout -k 10 250 python bench.py $B > gpurun_out/w17_p$n.json 2> gpurun_out/w17_p$n.err
done
python - <<'PY'
import json
for v in (0, 256, 512, 1024):
    try:
        d = json.load(open("gpurun_out/w17_p%d.json" % v))
        ks = {k["kernel"].split(" ", 1)[-1]: k for k in d["kernels"]}
        print(v, round(d["ms_per_step"], 1), d["finite"], [(n, round(ks[n]["ms"] / 2, 1), ks[n]["tflops"]) for n in ("conv_r00/conv_0", "conv_l01/conv_0", "conv_l01/conv_1", "conv_r1/conv_0", "conv_c/conv_0")])
    except Exception as e:
        print(v, "failed", e)
PY
