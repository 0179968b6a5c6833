#!/bin/bash
# tile order of conv_h3w_kernel: plane pairs fastest in blocks of NBE_WINO_ZBLOCK (0 = all pairs of the launch)
B="--no-strict --no-host-path --no-cpu-baseline"
NBE_WINO_ZBLOCK=4 timeout -k 10 200 python -m pytest tests/test_gpu_layers.py -x -q -m gpu -k "gauged" 2>&1 | tail -1
for v in 0 1 2 4 8 16; do
  NBE_WINO_ZBLOCK=$v NBE_PROF_LAYERS=1 timeout -k 10 250 python bench.py $B > gpurun_out/w23_$v.json 2> gpurun_out/w23_$v.err
  python - $v <<'PY'
import json, sys
d = json.load(open("gpurun_out/w23_%s.json" % sys.argv[1]))
ks = {k["kernel"].split(" ", 1)[-1]: k for k in d["kernels"]}
print("zblock", sys.argv[1], round(d["ms_per_step"], 1), [(n, round(ks[n]["ms"] / 2, 1)) for n in ("conv_r00/conv_0", "conv_r00/conv_1", "conv_l01/conv_0", "conv_l01/conv_1")])
PY
done
