#!/bin/bash
# rocprofv3 passes over the default bench workload (run from the repository root inside one gpurun call):
#   tools/gpu/profile.sh TAG [stats] [traffic] [sq] [insts] [-- extra bench.py arguments]
# stats:   --kernel-trace --stats            -> gpurun_out/${TAG}_bench_kernel_stats.csv, ${TAG}_bench_under_rocprof.json
# traffic: --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, MI355X guide) -> ${TAG}_pmc_fetch_write.txt, traffic.json
# sq:      matrix-pipe busy, waits, LDS, clock -> ${TAG}_pmc_sq.txt
# insts:   instruction mix (VALU / MFMA / LDS / VMEM) -> appended to ${TAG}_pmc_sq.txt
# Counters are collected with --kernel-trace only (never with --sys-trace & co.), and the program after `--` is python3 itself.
set -o pipefail
TAG=${1:?tag}; shift
PASSES=""; EXTRA=""
while [ $# -gt 0 ]; do
    if [ "$1" = "--" ]; then shift; EXTRA="$*"; break; fi
    PASSES="$PASSES $1"; shift
done
[ -z "$PASSES" ] && PASSES="stats traffic sq"
R=$(pwd)
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-strict --no-host-path --no-small-configs $EXTRA"
cd /tmp
for p in $PASSES; do
  case $p in
    stats)
      rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- $B --steps 2 --warmup 1 > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> $R/gpurun_out/prof_stats.err
      echo "stats rc=$?"
      f=$(find $R/gpurun_out/prof_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/gpurun_out/${TAG}_bench_kernel_stats.csv
      head -8 $R/gpurun_out/${TAG}_bench_kernel_stats.csv | cut -c1-200 ;;
    traffic)
      rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_fetch -- $B --steps 1 --warmup 0 > $R/gpurun_out/${TAG}_bench_pmc_fetch.json 2> $R/gpurun_out/prof_fetch.err
      echo "fetch rc=$?"
      rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_write -- $B --steps 1 --warmup 0 > $R/gpurun_out/${TAG}_bench_pmc_write.json 2> $R/gpurun_out/prof_write.err
      echo "write rc=$?"
      (cd $R && python3 tools/pmc_traffic.py gpurun_out/prof_fetch gpurun_out/prof_write --json gpurun_out/traffic.json --bench-json gpurun_out/${TAG}_bench_pmc_fetch.json > gpurun_out/${TAG}_pmc_fetch_write.txt 2>&1; head -12 gpurun_out/${TAG}_pmc_fetch_write.txt) ;;
    sq)
      rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/prof_sq -- $B --steps 1 --warmup 0 > $R/gpurun_out/${TAG}_bench_pmc_sq.json 2> $R/gpurun_out/prof_sq.err
      echo "sq rc=$?"
      (cd $R && for k in conv_h up_h3 stem_h3; do python3 tools/pmc_summary.py gpurun_out/prof_sq $k; done > gpurun_out/${TAG}_pmc_sq.txt 2>&1; cat gpurun_out/${TAG}_pmc_sq.txt) ;;
    insts)
      rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/prof_sq2 -- $B --steps 1 --warmup 0 > $R/gpurun_out/${TAG}_bench_pmc_insts.json 2> $R/gpurun_out/prof_sq2.err
      echo "insts rc=$?"
      (cd $R && python3 tools/pmc_summary.py gpurun_out/prof_sq2 conv_h --raw >> gpurun_out/${TAG}_pmc_sq.txt 2>&1; tail -12 gpurun_out/${TAG}_pmc_sq.txt) ;;
  esac
done
cd $R
# keep the merge small: the raw traces stay on the box
rm -rf gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_sq gpurun_out/prof_sq2
