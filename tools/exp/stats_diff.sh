#!/bin/bash
# Per-kernel time of the sampling pass (or train step: MODE=train) under two environments, side by side.
#   bash tools/exp/stats_diff.sh "ENV_A=.." "ENV_B=.."
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
if [ "$MODE" = train ]; then ARGS="--only-train --no-cpu-baseline --no-roofline --steps 5 --warmup 2"; else ARGS="--mode sample --big-batch 0 --no-cpu-baseline --no-roofline --steps 3 --warmup 1"; fi
i=0
for v in "$@"; do
  rm -rf /tmp/sd$i; env $v timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/sd$i -o t --output-format csv -- python $R/bench.py $ARGS > /tmp/sd$i.log 2>&1
  i=$((i+1))
done
python - <<'PY'
import csv, glob, collections, re
def load(d):
    f = glob.glob(f'{d}/**/t_kernel_stats.csv', recursive=True)[0]
    out = {}
    for r in csv.DictReader(open(f)):
        k = re.sub(r'\(anonymous namespace\)::', '', r['Name']); k = re.sub(r'^void ', '', k); k = re.sub(r'\(.*', '', k)[:64]
        out[k] = (int(r['Calls']), float(r['TotalDurationNs']) / 1e6)
    return out
a, b = load('/tmp/sd0'), load('/tmp/sd1')
keys = sorted(set(a) | set(b), key=lambda k: -abs(a.get(k, (0, 0))[1] - b.get(k, (0, 0))[1]))
print(f'{"kernel":64s} {"calls A":>8s} {"ms A":>9s} {"calls B":>8s} {"ms B":>9s} {"B-A":>8s}')
for k in keys[:40]:
    ca, ta = a.get(k, (0, 0.0)); cb, tb = b.get(k, (0, 0.0))
    print(f'{k:64s} {ca:8d} {ta:9.2f} {cb:8d} {tb:9.2f} {tb - ta:8.2f}')
print('total', sum(v[1] for v in a.values()), sum(v[1] for v in b.values()))
PY
