#!/bin/bash
# round 2 profiles of the default workload: kernel-trace stats, FETCH_SIZE / WRITE_SIZE passes, SQ counters
set -o pipefail
R=$(pwd)
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-strict --no-host-path"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- $B --steps 2 --warmup 1 > $R/gpurun_out/r02_bench_under_rocprof.json 2> $R/gpurun_out/prof_stats.err
echo "stats rc=$?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_fetch -- $B --steps 1 --warmup 0 > $R/gpurun_out/r02_bench_pmc_fetch.json 2> $R/gpurun_out/prof_fetch.err
echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_write -- $B --steps 1 --warmup 0 > $R/gpurun_out/r02_bench_pmc_write.json 2> $R/gpurun_out/prof_write.err
echo "write rc=$?"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/prof_sq -- $B --steps 1 --warmup 0 > $R/gpurun_out/r02_bench_pmc_sq.json 2> $R/gpurun_out/prof_sq.err
echo "sq rc=$?"
cd $R
python3 tools/pmc_traffic.py gpurun_out/prof_fetch gpurun_out/prof_write --json gpurun_out/traffic.json --bench-json gpurun_out/r02_bench_pmc_fetch.json > gpurun_out/r02_pmc_fetch_write.txt 2>&1
python3 tools/pmc_summary.py gpurun_out/prof_sq conv_h3 > gpurun_out/r02_pmc_sq.txt 2>&1
python3 tools/pmc_summary.py gpurun_out/prof_sq up_h3 >> gpurun_out/r02_pmc_sq.txt 2>&1
python3 tools/pmc_summary.py gpurun_out/prof_sq stem_h3 >> gpurun_out/r02_pmc_sq.txt 2>&1
f=$(find gpurun_out/prof_stats -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r02_bench_kernel_stats.csv
head -12 gpurun_out/r02_pmc_fetch_write.txt; cat gpurun_out/r02_pmc_sq.txt; head -8 gpurun_out/r02_bench_kernel_stats.csv | cut -c1-200
# keep the merge small: the raw traces stay on the box
rm -rf gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_sq
