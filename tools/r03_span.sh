#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_gpu_tilebook.py tests/test_gpu_engine.py tests/test_gpu_spconv.py -x -q --timeout 180 --timeout-method thread > gpurun_out/r03_span_tests.log 2>&1 || { tail -40 gpurun_out/r03_span_tests.log; exit 1; }
tail -2 gpurun_out/r03_span_tests.log
for g in 1 0; do
  PVX_ENGINE_SPAN_BOOKS=$g timeout -k 10 240 python3 tools/conv_bench.py --batch 24 --passes 8 > gpurun_out/r03_conv_span$g.txt 2>&1 || { tail -30 gpurun_out/r03_conv_span$g.txt; exit 1; }
  echo "PVX_ENGINE_SPAN_BOOKS=$g"; grep -E "conv_ms|128x128_k27|64x128|other" gpurun_out/r03_conv_span$g.txt
done
bash tools/r03_ab.sh "PVX_ENGINE_SPAN_BOOKS=1" "PVX_ENGINE_SPAN_BOOKS=0"
