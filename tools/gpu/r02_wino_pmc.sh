#!/bin/bash
# SQ counters of the default workload with the Winograd-z kernel on: clock and matrix-pipe occupancy of conv_h3w_kernel beside conv_h3g_kernel
set -o pipefail
R=$(pwd)
export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-strict --no-host-path"
cd /tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/prof_sq -- $B --steps 1 --warmup 0 > $R/gpurun_out/w8_bench_pmc_sq.json 2> $R/gpurun_out/prof_sq.err
echo "sq rc=$?"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/prof_sq2 -- $B --steps 1 --warmup 0 > $R/gpurun_out/w8_bench_pmc_sq2.json 2> $R/gpurun_out/prof_sq2.err
echo "sq2 rc=$?"
cd $R
python3 tools/pmc_summary.py gpurun_out/prof_sq conv_h3 > gpurun_out/w8_pmc_sq.txt 2>&1
python3 - <<'PY' >> gpurun_out/w8_pmc_sq.txt 2>&1
import csv, glob, os
from collections import defaultdict
d = "gpurun_out/prof_sq2"
cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
acc = defaultdict(lambda: defaultdict(float))
for r in csv.DictReader(open(cc)):
    if "conv_h3" in r["Kernel_Name"]:
        acc[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items():
    print(k, {c: "%.3e" % x for c, x in v.items()})
PY
cat gpurun_out/w8_pmc_sq.txt
rm -rf gpurun_out/prof_sq gpurun_out/prof_sq2
