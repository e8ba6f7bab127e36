#!/bin/bash
# round-3 evidence on the GPU box.  usage: bash tools/r03_profile.sh [stats] [pmc] [sq] [small]   -> gpurun_out/r03/
#   stats: rocprofv3 kernel stats of the SERIAL B=16 bench (what roofline.basis says) and of the default overlapped bench
#   pmc:   three TCC passes (tools/pmc_passes.sh) -> HBM traffic per kernel
#   sq:    one SQ-counter pass over the serial bench (tools/pmc_sq.sh), B=16
#   small: B=1 / B=4 bench lines
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
for what in "$@"; do
case $what in
stats)
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial -o r -- python3 bench.py --cpu-frames 0 --no-overlap --steps 20 --warmup 5 > $O/bench_serial_b24.json 2> $O/bench_serial.err
  cp $(ls $O/prof_serial/*kernel_stats.csv | head -1) $O/kernel_stats_serial_b24.csv
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -o r -- python3 bench.py --cpu-frames 0 --steps 50 --warmup 10 > $O/bench_default_b24_profiled.json 2> $O/bench_default.err
  cp $(ls $O/prof_default/*kernel_stats.csv | head -1) $O/kernel_stats_overlapped_b24.csv
  rm -rf $O/prof_serial $O/prof_default
  python3 tools/prof.py $O 25 40 || true ;;
pmc)
  bash tools/pmc_passes.sh ${PMC_BATCH:-24} r03 > $O/pmc_passes.log 2>&1
  cp gpurun_out/pmc_r03.txt $O/pmc_traffic_b${PMC_BATCH:-24}.txt; cp gpurun_out/pmc_r03.json $O/pmc_traffic_b${PMC_BATCH:-24}.json
  tail -40 $O/pmc_traffic_b${PMC_BATCH:-24}.txt ;;
sq)
  BATCH=${PMC_BATCH:-24} bash tools/pmc_sq.sh r03 > $O/pmc_sq.log 2>&1
  cp gpurun_out/pmc_sq_r03.txt $O/pmc_sq.txt; tail -50 $O/pmc_sq.txt ;;
small)
  for b in 1 4; do timeout -k 10 200 python3 bench.py --cpu-frames 0 --batch $b > $O/bench_b$b.json 2> $O/bench_b$b.err; done ;;
esac
done
ls $O
