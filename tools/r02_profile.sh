#!/bin/bash
# round-2 evidence on the GPU box: (a) rocprofv3 kernel stats of the SERIAL B=16 bench (what roofline.basis says), (b) of the default
# overlapped bench, (c) PMC traffic passes, (d) B=1 / B=4 lines.  Everything lands under gpurun_out/r02/.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial -o r -- python3 bench.py --cpu-frames 0 --no-overlap --steps 20 --warmup 5 > $O/bench_serial_b16.json 2> $O/bench_serial.err
cp $(ls $O/prof_serial/*kernel_stats.csv | head -1) $O/kernel_stats_serial_b16.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -o r -- python3 bench.py --cpu-frames 0 --steps 50 --warmup 10 > $O/bench_default_b16_profiled.json 2> $O/bench_default.err
cp $(ls $O/prof_default/*kernel_stats.csv | head -1) $O/kernel_stats_overlapped_b16.csv
rm -rf $O/prof_serial $O/prof_default
bash tools/pmc_passes.sh 16 r02 > $O/pmc_passes.log 2>&1
cp gpurun_out/pmc_r02.txt $O/pmc_traffic_b16.txt; cp gpurun_out/pmc_r02.json $O/pmc_traffic_b16.json
for b in 1 4; do timeout -k 10 200 python3 bench.py --cpu-frames 0 --batch $b --pool 16 > $O/bench_b$b.json 2> $O/bench_b$b.err; done
ls $O
