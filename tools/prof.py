"""print per-kernel averages of a rocprofv3 kernel_stats csv (helper for the profiles/ summaries)"""
import csv, glob, sys
path = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[-1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(path)))
tot = 0.0
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 26]:
    name = r['Name'].replace('(anonymous namespace)::', '').replace('_ZN12_GLOBAL__N_1', '')
    per_step = float(r['TotalDurationNs']) / 1e3 / steps
    tot += per_step
    print("%-74s calls/step=%5.1f avg_us=%8.1f us/step=%8.1f" % (name[:74], float(r['Calls']) / steps, float(r['AverageNs']) / 1e3, per_step))
print("sum of listed: %.1f us/step" % tot)
