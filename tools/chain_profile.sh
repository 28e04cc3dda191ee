#!/bin/bash
# bash tools/chain_profile.sh <tag> : marker-bracketed traces of the train step and the sampler + chain view
TAG=${1:-c}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/chain_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $O/train -o train --output-format csv -- python $R/bench.py --only-train --mark --no-cpu-baseline --no-roofline --steps 5 --warmup 2 > $O/train.log 2>&1
python $R/tools/trace_chain.py $O/train/train_kernel_trace.csv 5 > $O/train_chain.txt 2>&1
timeout 600 rocprofv3 --kernel-trace -d $O/sample -o sample --output-format csv -- python $R/bench.py --mode sample --mark --big-batch 0 --no-cpu-baseline --no-roofline --steps 3 --warmup 1 > $O/sample.log 2>&1
python $R/tools/trace_chain.py $O/sample/sample_kernel_trace.csv 3 > $O/sample_chain.txt 2>&1
rm -rf $O/train/*trace.csv $O/sample/*trace.csv
cat $O/train_chain.txt; cat $O/sample_chain.txt
