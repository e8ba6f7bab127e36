#!/bin/bash
# training-step evidence for profiles/r02_train_*: rocprofv3 kernel-trace run of the config-4 step (2 frames x 1M points, SyncBN + DDP)
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/train
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/train/prof -o train -- python3 tools/bench_train.py --batch 2 --points 250000 --sweeps 4 --steps 5 --warmup 2 --sync-bn > gpurun_out/train/bench_train_profiled.json 2> gpurun_out/train/prof.err
find gpurun_out/train/prof -name '*kernel_stats.csv' -exec cp {} gpurun_out/train/kernel_stats_train.csv \;
find gpurun_out/train/prof -name '*kernel_trace.csv' -delete
head -30 gpurun_out/train/kernel_stats_train.csv
