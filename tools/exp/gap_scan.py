"""Idle gaps > 100 us between consecutive kernels of a rocprofv3 kernel trace (all queues merged)."""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:60], r.get('Queue_Id', '?')))
rows.sort()
end = rows[0][1]
t0 = rows[0][0]
prev = rows[0]
for r in rows[1:]:
    gap = (r[0] - end) / 1e3
    if gap > 100:
        print(f'{(r[0] - t0) / 1e6:10.3f} ms  gap {gap:8.1f} us   after {prev[2][:40]} (q{prev[3]})  before {r[2][:40]} (q{r[3]})')
    if r[1] > end:
        end = r[1]
        prev = r
