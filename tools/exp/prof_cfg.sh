#!/bin/bash
# per-kernel breakdown of one train step of a bench configuration: bash tools/exp/prof_cfg.sh <config> [steps]
CFG=${1:-movie15x6}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$CFG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/train -o train --output-format csv -- python $R/bench.py --config $CFG --only-train --no-cpu-baseline --no-roofline --steps ${2:-4} --warmup 2 > $O/train_trace.log 2>&1
python $R/tools/trace_step.py $O/train/train_kernel_trace.csv 60 > $O/train_step_breakdown.txt 2>&1
rm -rf $O/train/*trace.csv
tail -1 $O/train_trace.log | cut -c1-300
