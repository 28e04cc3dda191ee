"""Aggregate rocprofv3 counter_collection CSVs (one or more passes) per kernel:
sum of each counter over the kernel's dispatches, dispatch count, per-launch averages."""
import collections
import csv
import glob
import re
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(set))
for pat in sys.argv[1:]:
    for path in glob.glob(pat, recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                k = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
                k = re.sub(r'^void ', '', k)
                k = re.sub(r'\(.*', '', k)[:70]
                agg[k][r['Counter_Name']] += float(r['Counter_Value'])
                cnt[k][r['Counter_Name']].add(r['Dispatch_Id'])
names = sorted({c for v in agg.values() for c in v})
print('kernel,dispatches,' + ','.join(names))
rows = []
for k, v in agg.items():
    n = max(len(s) for s in cnt[k].values())
    rows.append((v.get('GRBM_GUI_ACTIVE', 0.0), k, n, v))
for _, k, n, v in sorted(rows, reverse=True):
    print(f'"{k}",{n},' + ','.join(f'{v.get(c, 0.0):.6g}' for c in names))
