#!/bin/bash
# three rocprofv3 --pmc passes (one counter group each) of the default bench command, then tools/pmc_traffic.py
# usage (on the GPU box): bash tools/pmc_passes.sh <batch> <tag>   -> gpurun_out/pmc_<tag>.{txt,json}
set -e
B=${1:-16}; TAG=${2:-x}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for grp in "FETCH_SIZE:fetch" "WRITE_SIZE:write" "TCC_HIT_sum TCC_MISS_sum:l2"; do
  ctr=${grp%%:*}; name=${grp##*:}
  rm -rf gpurun_out/pmc_$name
  timeout -s KILL 240 rocprofv3 --pmc $ctr --output-format csv -d gpurun_out/pmc_$name -o p -- python3 bench.py --cpu-frames 0 --batch $B --pool $B --steps 3 --warmup 1 --profile-passes 1 > gpurun_out/pmc_$name.out 2> gpurun_out/pmc_$name.err
  echo "pass $name done"
done
python3 tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_l2 gpurun_out/pmc_$TAG.txt gpurun_out/pmc_$TAG.json
