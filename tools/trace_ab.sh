#!/bin/bash
# kernel-trace timelines of the default bench with the fused and the torch NMS front end (scheduling study)
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/trace_ab
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/fused -o t -- python3 bench.py --cpu-frames 0 --steps 20 --warmup 10 > $O/fused.json 2> $O/fused.err
PVX_NMS_TORCH_FRONT=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/torch -o t -- python3 bench.py --cpu-frames 0 --steps 20 --warmup 10 > $O/torch.json 2> $O/torch.err
ls -la $O/*/
