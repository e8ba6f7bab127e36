#!/bin/bash
# round 3: what the far (outside-the-window) gathers cost the slab kernels: default library vs the -DPVX_SLAB_ABL_NOFALLBACK build
set -e
mkdir -p gpurun_out
timeout -k 10 240 python3 tools/conv_bench.py --batch 16 --passes 8 --tag base --json gpurun_out/r03_conv_base.json > gpurun_out/r03_conv_base.txt 2>&1
PVX_LIB=$GRAFT_REPO_ROOT/openpcdet_amd/lib/libpcdet_abl_nofb.so timeout -k 10 240 python3 tools/conv_bench.py --batch 16 --passes 8 --tag nofb --json gpurun_out/r03_conv_nofb.json > gpurun_out/r03_conv_nofb.txt 2>&1
cat gpurun_out/r03_conv_base.txt gpurun_out/r03_conv_nofb.txt
