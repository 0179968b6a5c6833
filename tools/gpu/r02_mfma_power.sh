#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 120 tools/micro/mfma_power.bin > gpurun_out/r02_mfma_power.txt 2>&1
cat gpurun_out/r02_mfma_power.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -d $GRAFT_REPO_ROOT/gpurun_out/mfma_power_pmc -o pmc --output-format csv -- $GRAFT_REPO_ROOT/tools/micro/mfma_power.bin > /dev/null 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/mfma_power_pmc
