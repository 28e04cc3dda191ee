#!/bin/bash
# Kernel breakdown of the replayed train step (and optionally the sampling pass) under the given environment:
#   bash tools/exp/step_trace.sh tag [ENV=VAL ...]   -> gpurun_out/step_<tag>.txt (+ eval_<tag>.txt with SAMPLE=1)
TAG=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
env "$@" timeout 600 rocprofv3 --kernel-trace --stats -d $O/train -o train --output-format csv -- python $R/bench.py --only-train --no-cpu-baseline --no-roofline --steps 5 --warmup 2 > $O/train.log 2>&1
python $R/tools/trace_step.py $O/train/train_kernel_trace.csv 70 > $R/gpurun_out/step_$TAG.txt 2>&1
if [ -n "$SAMPLE" ]; then
  env "$@" timeout 600 rocprofv3 --kernel-trace --stats -d $O/sample -o sample --output-format csv -- python $R/bench.py --mode sample --big-batch 0 --no-cpu-baseline --no-roofline --steps 3 --warmup 1 > $O/sample.log 2>&1
  python $R/tools/trace_eval.py $O/sample/sample_kernel_trace.csv > $R/gpurun_out/eval_$TAG.txt 2>&1
fi
rm -rf $O
