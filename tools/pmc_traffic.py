"""rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum TCC_MISS_sum, one run each, csv) -> per-kernel HBM traffic per launch.
usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> <l2_dir> <out.txt> <out.json>
FETCH_SIZE is reported in KiB and, on gfx950, at half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM): it is doubled
here; the correction is calibrated for 16-B-per-lane streaming reads, so for 32..64-B gathered rows it is an upper bound."""
import collections
import csv
import glob
import json
import sys


def load(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for r in csv.DictReader(open(glob.glob(d + '/*counter_collection.csv')[0])):
        n = r['Kernel_Name']
        if 'sparse_conv_w32' in n or 'sparse_conv_slab' in n:
            kind = 'sparse_conv_slab' if 'sparse_conv_slab' in n else 'sparse_conv_w32'
            a = n.split('kernelIDF16b')[1] if 'kernelIDF16b' in n else n
            nums = [t for t in a.replace('E', ' ').replace('Li', ' ').split() if t.isdigit()]
            # (rocprofv3 demangles some instantiations only partly and drops the first template integer: cout then reads as the
            # window size; such rows are labelled by what is certain)
            key = '%s<cin=%s,cout=%s>' % (kind, nums[0], nums[1]) if int(nums[1]) <= 512 else '%s<16-channel rows>' % kind
        elif 'sparse_conv_mfma2' in n:
            key = 'sparse_conv_mfma2<cin=16,cout=%s>' % ('32' if 'ELi32E' in n else '16')
        else:
            key = n.replace('(anonymous namespace)::', '').split('(')[0][:48]
        agg[key][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[(key, r['Counter_Name'])] += 1
    return agg, cnt


f, fc = load(sys.argv[1]); w, wc = load(sys.argv[2]); l, lc = load(sys.argv[3])
rows = []
for k in f:
    n = fc[(k, 'FETCH_SIZE')]
    fe = f[k]['FETCH_SIZE'] / n * 1024 * 2
    wr = w[k]['WRITE_SIZE'] / max(wc[(k, 'WRITE_SIZE')], 1) * 1024 if k in w else 0.0
    h, m = (l[k]['TCC_HIT_sum'], l[k]['TCC_MISS_sum']) if k in l else (0, 0)
    rows.append(dict(kernel=k, launches=n, fetch_bytes_per_launch=fe, write_bytes_per_launch=wr, l2_hit=h / max(h + m, 1)))
rows.sort(key=lambda r: -(r['fetch_bytes_per_launch'] + r['write_bytes_per_launch']) * r['launches'])
with open(sys.argv[4], 'w') as o:
    o.write("%-46s %8s %14s %14s %7s\n" % ("kernel", "launches", "fetch MB/launch", "write MB/launch", "L2 hit"))
    for r in rows:
        o.write("%-46s %8d %14.1f %14.1f %7.3f\n" % (r['kernel'], r['launches'], r['fetch_bytes_per_launch'] / 1e6,
                                                     r['write_bytes_per_launch'] / 1e6, r['l2_hit']))
json.dump(rows, open(sys.argv[5], 'w'), indent=1)
print(open(sys.argv[4]).read())
