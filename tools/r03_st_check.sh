#!/bin/bash
# round 3: parity of the super-tile conv form + per-layer times (serial)
set -e
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_tilebook.py -x -q --timeout 120 --timeout-method thread > gpurun_out/r03_st_tests.log 2>&1 || { tail -30 gpurun_out/r03_st_tests.log; exit 1; }
tail -2 gpurun_out/r03_st_tests.log
timeout -k 10 240 python3 tools/conv_bench.py --batch 16 --passes 8 --tag st --json gpurun_out/r03_conv_st.json > gpurun_out/r03_conv_st.txt 2>&1 || { tail -30 gpurun_out/r03_conv_st.txt; exit 1; }
grep -E "conv_ms|64x64|128x128|32x32|other" gpurun_out/r03_conv_st.txt
