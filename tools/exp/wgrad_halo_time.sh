#!/bin/bash
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for h in 1 0; do
  rm -rf /tmp/wg$h; SDMI_WGRAD_HALO=$h timeout 300 rocprofv3 --kernel-trace -d /tmp/wg$h -o t --output-format csv -- python $R/tools/exp/wgrad_halo_time.py 2>/dev/null | grep "B="
  python - <<PY
import csv, glob, collections
rows = list(csv.DictReader(open(glob.glob('/tmp/wg$h/**/t_kernel_trace.csv', recursive=True)[0])))
rows = [r for r in rows if 'wgrad' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# five iterations per shape: print the median per (position in the shape's launch sequence)
out = [(r['Kernel_Name'].split('(')[0][-60:], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3) for r in rows]
per = len(out) // 25 if len(out) % 25 == 0 else None
print('HALO=$h kernels per call:', per)
if per:
    for s in range(5):
        for k in range(per):
            ts = sorted(out[(s * 5 + it) * per + k][1] for it in range(5))
            print(f'  shape {s}  {out[s * 5 * per + k][0]:60s} {ts[2]:8.1f} us')
PY
done
