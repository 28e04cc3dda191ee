#!/bin/bash
# bash tools/grid_profile.sh <tag> [filter]: per (kernel, grid) durations of the graphed sampler
TAG=${1:-g}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/grid_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $O/sample -o sample --output-format csv -- python $R/bench.py --mode sample $GRID_ARGS --mark --big-batch 0 --no-cpu-baseline --no-roofline --steps 3 --warmup 1 > $O/sample.log 2>&1
python $R/tools/trace_grid.py $O/sample/sample_kernel_trace.csv 3 "$2" > $O/sample_grid.txt 2>&1
rm -rf $O/sample/*trace.csv
cat $O/sample_grid.txt
