#!/bin/bash
# rocprofv3 --pmc passes over tools/conv_bench.py (per-layer serial launches), aggregated per conv kernel:  bash tools/r03_pmc_conv.sh <tag> [filter]
set -e
TAG=${1:-x}; FILTER=${2:-sparse_conv}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES:if" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE:sq"; do
  ctr=${grp%%:*}; name=${grp##*:}
  rm -rf gpurun_out/pmcc_$name
  timeout -s KILL 200 rocprofv3 --pmc $ctr --output-format csv -d gpurun_out/pmcc_$name -o p -- python3 tools/conv_bench.py --batch 16 --passes 2 > gpurun_out/pmcc_$name.out 2> gpurun_out/pmcc_$name.err
  python3 tools/pmc.py gpurun_out/pmcc_$name $FILTER > gpurun_out/pmcc_${name}_$TAG.txt
  rm -rf gpurun_out/pmcc_$name
done
cat gpurun_out/pmcc_if_$TAG.txt gpurun_out/pmcc_sq_$TAG.txt
