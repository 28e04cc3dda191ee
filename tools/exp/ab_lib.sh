#!/bin/bash
# Same-box A/B of two builds of libsdmi.so (SDMI_LIBPATH): sampling pass and train step, two runs each.
# usage: bash tools/exp/ab_lib.sh tools/exp/libsdmi_base.so [more env assignments for the B side]
cd $GRAFT_REPO_ROOT
BASE=$1; shift
one() {
  env "$@" timeout 300 python bench.py $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$MODE $*', round(d['ms_per_step'],3))"
}
for rep in 1 2; do
  MODE=sample; ARGS="--mode sample --big-batch 0 --no-cpu-baseline --no-roofline --steps 3 --warmup 1"
  one SDMI_LIBPATH=$BASE; one X=new "$@"
  MODE=train; ARGS="--only-train --no-cpu-baseline --no-roofline --steps 8 --warmup 3"
  one SDMI_LIBPATH=$BASE; one X=new "$@"
done
