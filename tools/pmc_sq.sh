#!/bin/bash
# one rocprofv3 --pmc pass of SQ stall / MFMA / LDS counters over the serial (--no-overlap) bench, aggregated per conv kernel
# usage (on the GPU box): bash tools/pmc_sq.sh <tag>   -> gpurun_out/pmc_sq_<tag>.txt
set -e
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_sq
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_sq -o p -- python3 bench.py --cpu-frames 0 --batch ${BATCH:-8} --pool ${BATCH:-8} --no-overlap --steps 3 --warmup 1 --profile-passes 1 > gpurun_out/pmc_sq.out 2> gpurun_out/pmc_sq.err
python3 tools/pmc.py gpurun_out/pmc_sq sparse_conv > gpurun_out/pmc_sq_$TAG.txt
tail -60 gpurun_out/pmc_sq_$TAG.txt
