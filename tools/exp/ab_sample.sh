#!/bin/bash
# A/B of sampling knobs on one GPU box: each argument one env assignment list (quoted), one bench.py run each.
cd $GRAFT_REPO_ROOT
for a in "$@"; do
  env $a timeout 300 python bench.py --mode sample --no-cpu-baseline --no-roofline --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sample $a', round(d['ms_per_step'],3), round(d['value']))"
done
