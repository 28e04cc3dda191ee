#!/bin/bash
# Pair-launch experiments on the GPU box: per-(kernel, grid) trace of the train step with sdmi_bwd_pair,
# then a sweep of the split policy knobs (each line: one bench.py run, same box).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-pair}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $O/tr -o t --output-format csv -- python $R/bench.py --only-train --mark --no-cpu-baseline --no-roofline --steps 4 --warmup 2 > $O/trace.log 2>&1
python $R/tools/trace_grid.py $O/tr/*/t_kernel_trace.csv 4 > $O/train_grid.txt 2>&1 || python $R/tools/trace_grid.py $O/tr/t_kernel_trace.csv 4 > $O/train_grid.txt 2>&1
python $R/tools/trace_step.py $(ls $O/tr/*/t_kernel_trace.csv $O/tr/t_kernel_trace.csv 2>/dev/null | head -1) 40 > $O/train_step.txt 2>&1
rm -rf $O/tr
cd $R
run() {
  env "$@" timeout 300 python bench.py --only-train --no-cpu-baseline --no-roofline --steps 8 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],3))"
}
run SDMI_BWD_PAIR=1
run SDMI_BWD_PAIR=0
run SDMI_PAIR_DGRAD=384
run SDMI_PAIR_DGRAD=192
run SDMI_PAIR_DGRAD=128
run SDMI_PAIR_SLOTS=384
run SDMI_PAIR_SLOTS=768
run SDMI_PAIR_MIN_STEPS=4
run SDMI_PAIR_MIN_STEPS=16
run SDMI_PAIR_MIN_STEPS=32
run SDMI_BWD_PAIR=1
