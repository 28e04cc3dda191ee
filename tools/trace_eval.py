"""Launch list of ONE UNet evaluation of the graph-replayed sampling pass, from a rocprofv3 kernel-trace CSV: the
kernels between two consecutive VQ launches (the sampler quantises every predicted x0: one vq_kernel per evaluation),
in launch order with duration / workgroups, then aggregated by kernel.  usage: python tools/trace_eval.py trace.csv [k]"""
import collections
import csv
import re
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        wgs = 1
        for a in 'XYZ':
            wgs *= max(int(r['Grid_Size_' + a]) // max(int(r['Workgroup_Size_' + a]), 1), 1)
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], wgs,
                     int(r['Workgroup_Size_X'])))
rows.sort()
marks = [i for i, r in enumerate(rows) if '::vq_' in r[2]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else -8
seg = rows[marks[k] + 1:marks[k + 1] + 1]
span = seg[-1][1] - seg[0][0]
print(f'kernels {len(seg)}  span {span / 1e3:.1f} us  summed {sum(e - s for s, e, *_ in seg) / 1e3:.1f} us')


def short(k):
    k = re.sub(r'\(anonymous namespace\)::', '', k)
    k = re.sub(r'^void ', '', k)
    return re.sub(r'\((Sdmi|int|float|unsigned|void|__hip|c10|at::|const).*', '', k)[:90]


agg = collections.defaultdict(lambda: [0, 0])
for s, e, k, g, w in seg:
    print(f'{(s - seg[0][0]) / 1e3:9.1f} +{(e - s) / 1e3:7.1f} us  wgs {g:6d} x{w:5d}  {short(k)}')
    agg[short(k)][0] += e - s
    agg[short(k)][1] += 1
print()
for k, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f'{t / 1e3:9.1f} us {c:4d} {t / c / 1e3:8.1f} us  {k}')
