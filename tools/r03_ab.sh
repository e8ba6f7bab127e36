#!/bin/bash
# A/B of engine switches: usage bash tools/r03_ab.sh "ENV1=a ENV2=b" "ENV1=c" ...   (each argument = one configuration, run twice)
mkdir -p gpurun_out
i=0
for cfg in "$@"; do
  for rep in 1 2; do
    i=$((i+1))
    env $cfg timeout -k 10 300 python3 bench.py --cpu-frames 0 > gpurun_out/ab_$i.json 2> gpurun_out/ab_$i.err || { echo "FAILED: $cfg"; tail -5 gpurun_out/ab_$i.err; exit 1; }
    python3 -c "
import json,sys
d=json.loads(open('gpurun_out/ab_$i.json').read().strip().splitlines()[-1]); print('$cfg |', d['value'], d['ms_per_step'], d['stages'])"
  done
done
