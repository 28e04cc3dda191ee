"""Summarise a rocprofv3 kernel-trace CSV: busy time (union of kernel intervals), summed kernel
durations and wall span of the last N-th fraction -- shows how much of a step is launch gaps."""
import csv
import sys

path = sys.argv[1]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
rows = rows[int(len(rows) * (1 - frac)):]
span = rows[-1][1] - rows[0][0]
summed = sum(e - s for s, e, _ in rows)
busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
for s, e, _ in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f'kernels={len(rows)} span={span / 1e6:.3f} ms  busy(union)={busy / 1e6:.3f} ms  '
      f'summed={summed / 1e6:.3f} ms  idle={100 * (1 - busy / span):.1f}%')
gaps = sorted(((rows[i + 1][0] - rows[i][1]) for i in range(len(rows) - 1)))
n = len(gaps)
print('gap percentiles us: p10 %.2f p50 %.2f p90 %.2f p99 %.2f' % tuple(
    gaps[int(n * q)] / 1e3 for q in (0.1, 0.5, 0.9, 0.99)))
