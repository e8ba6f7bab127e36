#!/bin/bash
# round 3: the whole GPU suite, then the default bench line
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/r03_gpu_tests.log 2>&1 || { tail -40 gpurun_out/r03_gpu_tests.log; exit 1; }
tail -3 gpurun_out/r03_gpu_tests.log
timeout -k 10 400 python3 bench.py --cpu-frames 0 > gpurun_out/r03_bench_nocpu.json 2> gpurun_out/r03_bench_nocpu.err || { tail -20 gpurun_out/r03_bench_nocpu.err; exit 1; }
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_bench_nocpu.json").read().strip().splitlines()[-1])
print("frames/s", d["value"], "ms/step", d["ms_per_step"], "roofline frac", d["roofline"]["frac"], "checked", d.get("checked"))
PY
