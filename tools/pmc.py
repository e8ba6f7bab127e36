"""aggregate a rocprofv3 --pmc counter_collection csv per kernel name (helper for profiles/)"""
import csv, glob, sys, collections
path = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))[-1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(path)):
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('_ZN12_GLOBAL__N_1', '')[:70]
    agg[name][r['Counter_Name']] += float(r['Counter_Value'])
    cnt[(name, r['Counter_Name'])] += 1
filt = sys.argv[2] if len(sys.argv) > 2 else ''
for name, d in agg.items():
    if filt and filt not in name: continue
    print(name)
    for c, v in sorted(d.items()):
        n = cnt[(name, c)]
        print("    %-28s total=%.4g  per_dispatch=%.4g (n=%d)" % (c, v, v / n, n))
