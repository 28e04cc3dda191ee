#!/bin/bash
# same-box A/B of the video train step (bench.py --config movie15x6): one run per argument (env assignments)
cd $GRAFT_REPO_ROOT
for a in "$@"; do
  env $a timeout 400 python bench.py --config movie15x6 --only-train --no-cpu-baseline --no-roofline --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('video $a', round(d['ms_per_step'],3), round(d['value']))"
done
