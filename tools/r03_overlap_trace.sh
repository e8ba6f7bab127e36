#!/bin/bash
# kernel-trace timeline of the default bench -> which kernels run beside each conv launch (tools/probes/overlap_from_trace.py)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/overlap_trace
rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/t -o t -- python3 bench.py --cpu-frames 0 --steps 20 --warmup 10 --profile-passes 1 > $O/bench.json 2> $O/bench.err
f=$(ls $O/t/*kernel_trace.csv | head -1)
python3 tools/probes/overlap_from_trace.py $f > $O/overlap.txt
cat $O/overlap.txt
rm -rf $O/t
