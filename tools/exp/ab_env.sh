#!/bin/bash
# Same-box A/B of environment settings: sampling pass and train step, two runs each.
# usage: bash tools/exp/ab_env.sh "A=1 B=2" "A=0" ...      (each argument one variant: space-separated assignments)
cd $GRAFT_REPO_ROOT
one() {
  env $2 timeout 300 python bench.py $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 [$2]', round(d['ms_per_step'],3))"
}
for rep in 1 2; do
  for v in "$@"; do
    ARGS="--mode sample --big-batch 0 --no-cpu-baseline --no-roofline --steps 3 --warmup 1"; one sample "$v"
    ARGS="--only-train --no-cpu-baseline --no-roofline --steps 8 --warmup 3"; one train "$v"
  done
done
