#!/bin/bash
# round 2: graph replay + address-bit test, whole GPU suite, bench with and without graphs
set -o pipefail
mkdir -p gpurun_out
python -m pytest "tests/test_gpu_api.py::test_graph_replay_is_identical" "tests/test_gpu_layers.py::test_dma_addressing_with_bit31_of_the_address_set" -x -q -s -m gpu > gpurun_out/r02_graph.log 2>&1 && \
python -m pytest tests -x -q -m gpu > gpurun_out/r02_gpu_all.log 2>&1 && \
python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-strict > gpurun_out/r02_bench_graph.json 2> gpurun_out/r02_bench_graph.err && \
NBE_GRAPH=0 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-strict --no-host-path > gpurun_out/r02_bench_nograph.json 2> gpurun_out/r02_bench_nograph.err
rc=$?
tail -15 gpurun_out/r02_graph.log; tail -3 gpurun_out/r02_gpu_all.log
for f in graph nograph; do python - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/r02_bench_$f.json") if l.startswith("{")][-1])
    print("$f", round(d["value"]/1e6,2), "Mvox/s", round(d["ms_per_step"],1), "ms", d["roofline"]["kernel"], round(d["roofline"]["frac"],4), d.get("host_path"))
except Exception as e: print("$f", e)
PY
done
exit $rc
