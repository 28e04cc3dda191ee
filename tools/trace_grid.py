"""Per (kernel template, grid) durations inside the marker-bracketed region of a kernel trace.
usage: python tools/trace_grid.py <kernel_trace.csv> <steps> [name filter]"""
import collections
import csv
import re
import sys

MARK = 'sqerr_rows_kernel'
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'],
                     r.get('Grid_Size', r.get('Grid_Size_X', '?')), r.get('Workgroup_Size', r.get('Workgroup_Size_X', '?'))))
rows.sort()
steps = int(sys.argv[2])
flt = sys.argv[3] if len(sys.argv) > 3 else ''
marks = [i for i, r in enumerate(rows) if MARK in r[2]]
seg = rows[marks[-2] + 1:marks[-1]]
agg = collections.defaultdict(lambda: [0, 0.0])
for s, e, k, g, w in seg:
    k = re.sub(r'\(anonymous namespace\)::', '', k)
    k = re.sub(r'^void ', '', k)
    k = re.sub(r'\(.*', '', k)
    if flt in k:
        a = agg[(k[:70], g, w)]
        a[0] += 1
        a[1] += (e - s) / 1e3
for (k, g, w), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f'{t / steps:9.1f} us/step {n / steps:7.1f} calls  {t / n:7.2f} us avg  grid {g:>9} wg {w:>4}  {k}')
