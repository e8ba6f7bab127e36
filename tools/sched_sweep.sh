#!/bin/bash
# scheduling sweep of the default bench: where the NMS launches go (stream, order, conv layer they wait for)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/sweep
run() { timeout -k 10 200 python -u bench.py --cpu-frames 0 --profile-passes 1 "$@" > gpurun_out/sweep/o.json || exit 1
        python -c "
import json,sys;b=json.load(open('gpurun_out/sweep/o.json'));print('%8.1f fps %.3f ms  %s'%(b['value'],b['ms_per_step'],' '.join(sys.argv[1:])))" -- "$@" | tee -a gpurun_out/sweep/results.txt; }
for a in "$@"; do run $a || exit 1; done
