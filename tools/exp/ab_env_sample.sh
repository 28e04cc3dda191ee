#!/bin/bash
# Same-box A/B of environment settings on ONE leg (MODE=sample | train), rotating the order of the variants between
# repetitions so that a drift of the box (clock / temperature) does not line up with one variant.
cd $GRAFT_REPO_ROOT
if [ "$MODE" = train ]; then ARGS="--only-train --no-cpu-baseline --no-roofline --steps 8 --warmup 3"; else ARGS="--mode sample --big-batch 0 --no-cpu-baseline --no-roofline --steps 3 --warmup 1"; fi
vars=("$@"); n=${#vars[@]}
for rep in 0 1 2; do
  for i in $(seq 0 $((n-1))); do
    v="${vars[$(( (i + rep) % n ))]}"
    env $v timeout 300 python bench.py $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${MODE:-sample} rep$rep [$v]', round(d['ms_per_step'],3))"
  done
done
