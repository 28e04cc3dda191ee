#!/bin/bash
# Round profile collection on the GPU box (outputs under gpurun_out/prof_<tag>/, copy the summaries
# into profiles/).  usage: bash tools/collect_profiles.sh r03
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. kernel-trace + stats of the bench command (graph replay), train and sampling
timeout 600 rocprofv3 --kernel-trace --stats -d $O/train -o train --output-format csv -- python $R/bench.py --only-train --no-cpu-baseline --no-roofline --steps 5 --warmup 2 > $O/train_trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/sample -o sample --output-format csv -- python $R/bench.py --mode sample --big-batch 0 --no-cpu-baseline --no-roofline --steps 3 --warmup 1 > $O/sample_trace.log 2>&1
# 2. (the per-kernel PMC tables come from bench.py --pmc-table: tools/collect_bench.sh)
# 3. the fp8 sampling configuration and the COCO-224 / DINO configuration (kernel stats only)
timeout 600 rocprofv3 --kernel-trace --stats -d $O/sample_fp8 -o sample_fp8 --output-format csv -- python $R/bench.py --mode sample --dtype fp8 --big-batch 0 --no-cpu-baseline --no-roofline --steps 3 --warmup 1 > $O/sample_fp8_trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/coco -o coco --output-format csv -- python $R/bench.py --config coco224 --mode sample --dtype fp8 --no-cpu-baseline --no-roofline --steps 2 --warmup 1 > $O/coco_trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/coco_train -o coco_train --output-format csv -- python $R/bench.py --config coco224 --mode train --only-train --no-cpu-baseline --no-roofline --steps 4 --warmup 2 > $O/coco_train_trace.log 2>&1
cd $R
cp $O/coco_train/coco_train_kernel_stats.csv $O/${TAG}_coco224_train_b16_bf16_kernel_stats.csv 2>/dev/null
cp $O/sample_fp8/sample_fp8_kernel_stats.csv $O/${TAG}_sample_b64_fp8_kernel_stats.csv 2>/dev/null
cp $O/coco/coco_kernel_stats.csv $O/${TAG}_coco224_sample_b16_fp8_kernel_stats.csv 2>/dev/null
cp $O/train/train_kernel_stats.csv $O/${TAG}_train_b64_bf16_kernel_stats.csv 2>/dev/null
cp $O/sample/sample_kernel_stats.csv $O/${TAG}_sample_b64_bf16_kernel_stats.csv 2>/dev/null
python tools/trace_step.py $O/train/train_kernel_trace.csv 60 > $O/${TAG}_train_step_breakdown.txt 2>&1
python tools/trace_eval.py $O/sample/sample_kernel_trace.csv > $O/${TAG}_sample_eval_launches.txt 2>&1
rm -rf $O/train/*trace.csv $O/sample/*trace.csv $O/sample_fp8/*trace.csv $O/coco/*trace.csv $O/coco_train/*trace.csv
ls -la $O | head -40
