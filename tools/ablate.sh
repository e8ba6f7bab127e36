#!/bin/bash
# timing-only ablations of the sparse-conv kernel (see PVX_ABLATE in csrc/sparse_conv.hip), kernel time from rocprofv3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "0 0" "1 0" "2 0" "3 0" "7 0" "16 0" "31 1"; do
  set -- $cfg
  rm -rf gpurun_out/abl
  PVX_ABLATE=$1 PVX_ABLATE_GATHER=$2 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl -- python3 bench.py --steps 10 --warmup 3 --cpu-frames 0 --profile-passes 1 > /dev/null 2>&1
  echo -n "ABLATE=$1 g=$2: "
  python3 - <<'PY'
import csv, glob, re
path = sorted(glob.glob("gpurun_out/abl/**/*kernel_stats.csv", recursive=True))[-1]
out = []
for r in csv.DictReader(open(path)):
    n = r['Name']
    if 'mfma2' not in n: continue
    m = re.search(r'Li(\d+)ELi(\d+)ELi(\d)ELi(\d+)ELb(\d)', n)
    tag = "%sx%s" % (m.group(1), m.group(2)) if m else "16x?"
    out.append((tag, float(r['AverageNs']) / 1e3))
print("  ".join("%s:%.0f" % t for t in sorted(out)))
PY
done
