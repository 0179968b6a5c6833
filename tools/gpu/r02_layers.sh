#!/bin/bash
# per-layer times of the default workload (NBE_PROF_LAYERS=1: one profile entry per layer)
NBE_PROF_LAYERS=1 timeout -k 10 250 python bench.py --no-strict --no-host-path --no-cpu-baseline > gpurun_out/layers_on.json 2> gpurun_out/layers_on.err
NBE_WINO=0 NBE_PROF_LAYERS=1 timeout -k 10 250 python bench.py --no-strict --no-host-path --no-cpu-baseline > gpurun_out/layers_off.json 2> gpurun_out/layers_off.err
python - <<'PY'
import json
on = {k["kernel"].split(" ", 1)[-1]: k for k in json.load(open("gpurun_out/layers_on.json"))["kernels"]}
off = {k["kernel"].split(" ", 1)[-1]: k for k in json.load(open("gpurun_out/layers_off.json"))["kernels"]}
print("%-22s %8s %8s %7s %7s %6s" % ("layer", "ms on", "ms off", "TF on", "TF off", "ratio"))
for name, k in sorted(on.items(), key=lambda kv: -kv[1]["ms"]):
    o = off.get(name)
    if o:
        print("%-22s %8.1f %8.1f %7.1f %7.1f %6.3f  %s" % (name, k["ms"] / 2, o["ms"] / 2, k["tflops"] or 0, o["tflops"] or 0, o["ms"] / k["ms"], k["kernel"].split(" ")[0]))
PY
