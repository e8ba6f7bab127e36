#!/bin/bash
# round 3 iteration loop for the wave-specialised conv kernel: parity (tile book tests), per-layer times, stamps of the diagnostic build
set -e
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_tilebook.py -x -q --timeout 120 --timeout-method thread > gpurun_out/r03_st_tests.log 2>&1 || { tail -30 gpurun_out/r03_st_tests.log; exit 1; }
tail -1 gpurun_out/r03_st_tests.log
timeout -k 10 240 python3 tools/conv_bench.py --batch 16 --passes 8 --tag ws --json gpurun_out/r03_conv_ws.json > gpurun_out/r03_conv_ws.txt 2>&1 || { tail -30 gpurun_out/r03_conv_ws.txt; exit 1; }
grep -E "conv_ms|64x64|128x128" gpurun_out/r03_conv_ws.txt
if [ -f openpcdet_amd/lib/libpcdet_stamps.so ]; then
  PVX_LIB=$GRAFT_REPO_ROOT/openpcdet_amd/lib/libpcdet_stamps.so ST_ONLY=${ST_ONLY:-64x64} timeout -k 10 200 python3 tools/st_stamps.py 2>&1 | grep -v amdgpu.ids | tail -${ST_TAIL:-14}
fi
