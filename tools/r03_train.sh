#!/bin/bash
# round-3 training-step lines (BASELINE.json configs[3], one GPU): plain / SyncBN + DDP, fused / foreach AdamW; optional rocprofv3 stats of the first
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_train; mkdir -p $O
export TMPDIR=/tmp
run() { name=$1; shift; timeout -k 10 300 python3 tools/bench_train.py --batch 2 --points 250000 --sweeps 4 --steps 10 --warmup 3 "$@" > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; exit 1; }
        python3 -c "import json; d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('$name', d['ms_per_step'], 'ms', d['value'], 'frames/s')"; }
run plain --no-ddp
run plain_foreach --no-ddp --no-fused-opt
run ddp_syncbn --sync-bn
if [ "$1" = "prof" ]; then
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o train -- python3 tools/bench_train.py --batch 2 --points 250000 --sweeps 4 --steps 5 --warmup 2 --no-ddp > $O/plain_profiled.json 2> $O/prof.err
  find $O/prof -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats_plain.csv \;
  rm -rf $O/prof
  head -25 $O/kernel_stats_plain.csv | cut -c1-150
fi
