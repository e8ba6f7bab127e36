"""From a rocprofv3 --kernel-trace CSV of bench.py: for every conv kernel family, the launches INSIDE the timed region, their mean
duration, and which other kernels ran at the same time (share of the conv launch's duration covered by each co-runner)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
def short(n):
    for key, s in (("slab_roll", "roll32"), ("slab_ws", "ws64"), ("slab_stream_kernelIDF16bLi128ELi128", "stream128"), ("slab_stream_kernelIDF16bLi64ELi128", "stream64_128"),
                   ("slab_kernelIDF16bLi32ELi64", "slab32_64"), ("sparse_conv_slab_tb", "tb16"), ("sparse_conv_w32", "w32_k3")):
        if key in n: return s
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0].split("<")[0][:28]
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in rows]
ev.sort()
convs = ("roll32", "ws64", "stream128", "stream64_128", "slab32_64", "tb16", "w32_k3")
# timed region ~ the middle 60 % of the ws64 launches
ws = [e for e in ev if e[2] == "ws64"]
lo, hi = ws[len(ws) // 5][0], ws[len(ws) * 4 // 5][1]
others = [e for e in ev if e[2] not in convs and e[1] > lo and e[0] < hi]
for fam in convs:
    L = [e for e in ev if e[2] == fam and e[0] >= lo and e[1] <= hi]
    if not L: continue
    cov = collections.Counter()
    alone = 0.0
    for s, e, _ in L:
        marks = []
        for os_, oe, on in others:
            if oe <= s or os_ >= e: continue
            a, b = max(s, os_), min(e, oe)
            cov[on] += (b - a)
            marks.append((a, b))
        marks.sort()
        covered, cur = 0, s
        for a, b in marks:
            if b > cur: covered += b - max(a, cur); cur = b
        alone += (e - s) - covered
    tot = sum(e - s for s, e, _ in L)
    print("%-13s %3d launches, mean %.1f us, nothing beside it %.0f %% of the time; beside it (share of its time): %s"
          % (fam, len(L), tot / len(L) / 1e3, 100 * alone / tot, ", ".join("%s %.0f%%" % (k, 100 * v / tot) for k, v in cov.most_common(7))))
