"""Critical-chain view of a rocprofv3 kernel-trace CSV (graph-replayed bench run): per queue, the
summed kernel durations, the summed gaps between consecutive kernels and the gap histogram inside the
region bracketed by the bench's marker kernels (bench.py --mark).
usage: python tools/trace_chain.py <kernel_trace.csv> <steps>"""
import collections
import csv
import re
import sys

MARK = 'sqerr_rows_kernel'
rows = []
with open(sys.argv[1]) as f:
    rd = csv.DictReader(f)
    for r in rd:
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'],
                     r.get('Queue_Id', '0'), r.get('Stream_Id', '0')))
steps = int(sys.argv[2])
rows.sort()


def base(k):
    k = re.sub(r'\(anonymous namespace\)::', '', k)
    k = re.sub(r'^void ', '', k)
    return re.sub(r'[<(].*', '', k).strip()


marks = [i for i, r in enumerate(rows) if base(r[2]) == MARK]
seg = rows[marks[-2] + 1:marks[-1]]
span = (seg[-1][1] - seg[0][0]) / 1e3 / steps
print(f'kernels/step {len(seg) / steps:.0f}  span/step {span:.1f} us')
byq = collections.defaultdict(list)
for r in seg:
    byq[(r[3], r[4])].append(r)
for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    dur = sum(e - s for s, e, *_ in rs) / 1e3 / steps
    gaps = [(rs[i + 1][0] - rs[i][1]) / 1e3 for i in range(len(rs) - 1)]
    small = [g for g in gaps if g < 50]
    print(f'queue {q}: {len(rs) / steps:.0f} kernels/step, kernel time {dur:.0f} us/step, '
          f'gaps<50us: sum {sum(small) / steps:.0f} us/step, median {sorted(small)[len(small) // 2] if small else 0:.2f} us, '
          f'p90 {sorted(small)[int(len(small) * .9)] if small else 0:.2f} us')
    if len(rs) > 0.3 * len(seg):
        # short kernels on the main queue
        bins = collections.Counter()
        tb = collections.Counter()
        for s, e, k, *_ in rs:
            d = (e - s) / 1e3
            b = '<5' if d < 5 else '<10' if d < 10 else '<20' if d < 20 else '<50' if d < 50 else '>=50'
            bins[b] += 1
            tb[b] += d
        for b in ('<5', '<10', '<20', '<50', '>=50'):
            print(f'    duration {b:>4} us: {bins[b] / steps:6.0f} kernels/step  {tb[b] / steps:8.0f} us/step')
        agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
        for i, (s, e, k, *_) in enumerate(rs):
            a = agg[base(k)]
            a[0] += 1
            a[1] += (e - s) / 1e3
            if i + 1 < len(rs):
                g = (rs[i + 1][0] - e) / 1e3
                a[2] += g if g < 50 else 0
        print('    per kernel on this queue: calls/step, us/step, following-gap us/step')
        for k, a in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))[:40]:
            print(f'    {a[0] / steps:6.0f} {a[1] / steps:8.0f} {a[2] / steps:8.0f}  {k}')
