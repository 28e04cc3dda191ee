#!/bin/bash
# A/B of train-step knobs on one GPU box: each line one bench.py run (ms per step).
cd $GRAFT_REPO_ROOT
run() {
  env "$@" timeout 300 python bench.py --only-train --no-cpu-baseline --no-roofline --steps 8 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],3))"
}
for a in "$@"; do run $a; done
