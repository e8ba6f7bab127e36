#!/bin/bash
# round 3: the whole GPU suite, then the bench line without and with the CPU baseline; optional: one high-priority run under the watchdog
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/r03_gpu_tests.log 2>&1 || { tail -40 gpurun_out/r03_gpu_tests.log; exit 1; }
tail -3 gpurun_out/r03_gpu_tests.log
timeout -k 10 400 python3 bench.py --cpu-frames 0 > gpurun_out/r03_bench_nocpu.json 2> gpurun_out/r03_bench_nocpu.err || { tail -20 gpurun_out/r03_bench_nocpu.err; exit 1; }
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_bench_nocpu.json").read().strip().splitlines()[-1])
print("frames/s", d["value"], "ms/step", d["ms_per_step"], "roofline frac", d["roofline"]["frac"], "overlapped", d["roofline"].get("frac_overlapped"), "checked", d.get("checked"))
PY
if [ "$1" = "cpu" ] || [ "$2" = "cpu" ]; then
  timeout -k 10 600 python3 bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err || { tail -20 gpurun_out/r03_bench.err; exit 1; }
  python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_bench.json").read().strip().splitlines()[-1])
print("frames/s", d["value"], "cpu", json.dumps(d["cpu_baseline"]))
PY
fi
if [ "$1" = "high" ] || [ "$2" = "high" ]; then
  # LAST step of the call on purpose: the round-2 hang was seen with high-priority geometry streams; the watchdog ends it with a report
  timeout -k 10 200 python3 bench.py --cpu-frames 0 --geo-priority high --watchdog-s 20 > gpurun_out/r03_bench_high.json 2> gpurun_out/r03_bench_high.err || { echo "high-priority run: exit $?"; tail -5 gpurun_out/r03_bench_high.err; tail -c 2000 gpurun_out/r03_bench_high.json; exit 0; }
  tail -c 600 gpurun_out/r03_bench_high.json
fi
