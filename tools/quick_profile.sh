#!/bin/bash
# Quick per-kernel breakdown of one graph-replayed train step (and, with "sample", of the sampler):
#   bash tools/quick_profile.sh <tag> [sample]
TAG=${1:-q}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/train -o train --output-format csv -- python $R/bench.py --only-train --no-cpu-baseline --no-roofline --steps 5 --warmup 2 > $O/train_trace.log 2>&1
python $R/tools/trace_step.py $O/train/train_kernel_trace.csv 70 > $O/train_step_breakdown.txt 2>&1
cp $O/train/train_kernel_stats.csv $O/train_kernel_stats.csv 2>/dev/null
if [ "$2" = "sample" ]; then
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/sample -o sample --output-format csv -- python $R/bench.py --mode sample --big-batch 0 --no-cpu-baseline --no-roofline --steps 3 --warmup 1 > $O/sample_trace.log 2>&1
  cp $O/sample/sample_kernel_stats.csv $O/sample_kernel_stats.csv 2>/dev/null
fi
rm -rf $O/train/*trace.csv $O/sample/*trace.csv
tail -2 $O/train_trace.log | cut -c1-400
head -75 $O/train_step_breakdown.txt
