#!/bin/bash
# one rocprofv3 --pmc pass of instruction-fetch counters over the serial bench, aggregated per conv kernel
set -e
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_if
timeout -s KILL 240 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_if -o p -- python3 bench.py --cpu-frames 0 --batch ${BATCH:-8} --pool ${BATCH:-8} --no-overlap --steps 3 --warmup 1 --profile-passes 1 > gpurun_out/pmc_if.out 2> gpurun_out/pmc_if.err
python3 tools/pmc.py gpurun_out/pmc_if sparse_conv > gpurun_out/pmc_if_$TAG.txt
