#!/bin/bash
# quick A/B on the GPU box: conv parity tests + one bench line with per-conv-shape kernel times (rocprofv3 kernel trace)
set -e
timeout -k 10 300 python -u -m pytest tests/test_gpu_spconv.py tests/test_gpu_engine.py -x -q --timeout 120 --timeout-method thread > gpurun_out/q_tests.log 2>&1 || { tail -20 gpurun_out/q_tests.log; exit 1; }
tail -1 gpurun_out/q_tests.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/q_prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/q_prof -o r -- python3 bench.py --cpu-frames 0 --batch 8 --pool 8 --no-overlap --steps 20 --warmup 5 > gpurun_out/q_bench.json 2> gpurun_out/q_bench.err
python3 - <<'PY'
import json, sqlite3
d = json.loads(open("gpurun_out/q_bench.json").read().strip().splitlines()[-1])
print("frames/s", d["value"], d["stages"])
db = sqlite3.connect("gpurun_out/q_prof/r_results.db")
rows = list(db.execute("select name, count(*), avg(end-start) from kernels where name like '%sparse_conv%' group by name order by 2*3 desc"))
for n, c, a in rows:
    short = n.split("kernelI")[1][:40] if "kernelI" in n else n[:60]
    print("  %-42s n=%4d avg %7.1f us" % (short, c, a / 1e3))
PY
