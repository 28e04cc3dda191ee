#!/bin/bash
# per-kernel times of the fused SpatialTransformer block (phase A / phase B) at B = 64
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/stprof; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/stprof -o st --output-format csv -- python tools/exp/st_check.py 64 > gpurun_out/stprof/log.txt 2>&1
grep "chain\|fused vs" gpurun_out/stprof/log.txt
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/stprof/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r['Name'][:90], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
