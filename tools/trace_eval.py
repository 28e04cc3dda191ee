"""Ordered launch list of ONE graph-replayed sampling pass (20 UNet evaluations), from a rocprofv3 kernel-trace CSV
(passes are delimited by the timestep-embedding kernel, which runs once per pass for all 20 steps).  usage: python tools/trace_eval.py trace.csv [which]"""
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
                     int(r['Workgroup_Size_X']), int(r['LDS_Block_Size'])))
rows.sort()
marks = [i for i, r in enumerate(rows) if 'time_emb_kernel' in r[2]]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -3
seg = rows[marks[which]:marks[which + 1]]
span = seg[-1][1] - seg[0][0]
print(f'kernels {len(seg)}  span {span / 1e3:.1f} us  summed {sum(e - s for s, e, *_ in seg) / 1e3:.1f} us')
prev_end = seg[0][0]
for s, e, k, g, w, lds in seg:
    k = re.sub(r'\(anonymous namespace\)::', '', k)
    k = re.sub(r'^void ', '', k)
    k = re.sub(r'\((Sdmi|int|float|unsigned|void|__hip|c10|at::|const).*', '', k)[:90]
    print(f'{(s - seg[0][0]) / 1e3:9.1f} +{(e - s) / 1e3:7.1f} us  gap {(s - prev_end) / 1e3:5.1f}  wgs {g:6d} x{w:5d} lds {lds // 1024:4d}K  {k}')
    prev_end = e
