"""Launch list of ONE graph-replayed train step in start order, from a rocprofv3 kernel-trace CSV (steps are
delimited by the Adam kernels): start offset, duration, workgroups, queue, kernel.  Overlapping launches (side streams)
show as starts before the previous end.  usage: python tools/trace_step_list.py trace.csv"""
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
                     int(r['Workgroup_Size_X']), r.get('Queue_Id', '?')))
rows.sort()
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r[2]]
ends = [adam[i] for i in range(len(adam)) if i + 1 == len(adam) or adam[i + 1] != adam[i] + 1]
# (bench.py runs one more step behind its closing barrier: that one starts on an idle queue, ~0.4 ms of host launch
# latency in front of its first kernel -- take the last step of the timed loop instead)
k = -2 if len(ends) >= 3 else -1
seg = rows[ends[k - 1] + 1:ends[k] + 1]


def short(k):
    k = re.sub(r'\(anonymous namespace\)::', '', k)
    k = re.sub(r'^void ', '', k)
    return re.sub(r'\((Sdmi|int|float|unsigned|void|__hip|c10|at::|const).*', '', k)[:80]


t0, prev_end = seg[0][0], seg[0][0]
for s, e, k, g, w, q in seg:
    gap = (s - prev_end) / 1e3
    print(f'{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:7.1f} us  gap {gap:6.1f}  wgs {g:6d} x{w:5d}  q{q}  {short(k)}')
    prev_end = max(prev_end, e)
