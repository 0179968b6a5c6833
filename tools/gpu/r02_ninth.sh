#!/bin/bash
# the N > 1 bench path on one card (gloo rig), graph A/B without profiling
set -o pipefail
mkdir -p gpurun_out
NBE_BENCH_ONE_GPU=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 1 --warmup 1 --size 256 --ndiv 2 --no-cpu-baseline > gpurun_out/r02_bench_2ranks_one_card.json 2> gpurun_out/r02_bench_2ranks_one_card.err
echo "2 ranks rc=$?"; tail -c 1500 gpurun_out/r02_bench_2ranks_one_card.json; tail -3 gpurun_out/r02_bench_2ranks_one_card.err
python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-strict --no-host-path --no-profile > gpurun_out/r02_bench_graph_on.json 2> gpurun_out/r02_bench_graph_on.err
NBE_GRAPH=0 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-strict --no-host-path --no-profile > gpurun_out/r02_bench_graph_off.json 2> gpurun_out/r02_bench_graph_off.err
for f in graph_on graph_off; do python - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/r02_bench_$f.json") if l.startswith("{")][-1])
    print("$f", round(d["value"]/1e6,2), "Mvox/s", round(d["ms_per_step"],2), "ms", d["config"].get("tiles_replayed_from_hipgraphs"))
except Exception as e: print("$f", e)
PY
done
