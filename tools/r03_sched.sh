#!/bin/bash
# round 3: tiles dealt by cost (pvx_tilebook_schedule) -- parity, per-layer serial times with and without, bench A/B
set -e
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_gpu_tilebook.py tests/test_gpu_engine.py -x -q --timeout 180 --timeout-method thread > gpurun_out/r03_sched_tests.log 2>&1 || { tail -40 gpurun_out/r03_sched_tests.log; exit 1; }
tail -2 gpurun_out/r03_sched_tests.log
for g in 1 0; do
  PVX_ENGINE_SCHEDULE=$g timeout -k 10 240 python3 tools/conv_bench.py --batch ${SCHED_BATCH:-24} --passes 8 --tag sched$g > gpurun_out/r03_conv_sched$g.txt 2>&1 || { tail -30 gpurun_out/r03_conv_sched$g.txt; exit 1; }
  echo "PVX_ENGINE_SCHEDULE=$g"; grep -E "conv_ms|64x64|other" gpurun_out/r03_conv_sched$g.txt | head -30
done
bash tools/r03_ab.sh "PVX_ENGINE_SCHEDULE=1" "PVX_ENGINE_SCHEDULE=0"
