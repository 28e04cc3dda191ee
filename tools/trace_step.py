"""Per-kernel breakdown of ONE graph-replayed train step from a rocprofv3 kernel-trace CSV
(steps are delimited by the Adam kernels)."""
import collections
import csv
import re
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r[2]]
# a step = (previous step's last adam, this step's last adam]
ends = [adam[i] for i in range(len(adam)) if i + 1 == len(adam) or adam[i + 1] != adam[i] + 1]
# (bench.py runs one more step behind its closing barrier: that one starts on an idle queue, ~0.4 ms of host launch
# latency in front of its first kernel -- take the last step of the timed loop instead)
k = -2 if len(ends) >= 3 else -1
seg = rows[ends[k - 1] + 1:ends[k] + 1]
span = seg[-1][1] - seg[0][0]
busy, cs, ce = 0, seg[0][0], seg[0][1]
for s, e, _ in sorted(seg):
    if s > ce:
        busy += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print(f'kernels {len(seg)}  span {span / 1e6:.3f} ms  busy(union) {busy / 1e6:.3f} ms  '
      f'summed {sum(e - s for s, e, _ in seg) / 1e6:.3f} ms')
agg = collections.defaultdict(lambda: [0, 0])
for s, e, k in seg:
    k = re.sub(r'\(anonymous namespace\)::', '', k)
    k = re.sub(r'^void ', '', k)
    k = re.sub(r'\(.*', '', k)[:80]
    agg[k][0] += e - s
    agg[k][1] += 1
for k, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(sys.argv[2]) if len(sys.argv) > 2 else 45]:
    print(f'{t / 1e6:8.3f} ms {c:5d} {t / c / 1e3:8.1f} us  {k}')
