#!/bin/bash
# Round bench lines on the GPU box (outputs under gpurun_out/bench_<tag>/, copy into profiles/).
# usage: bash tools/collect_bench.sh r03
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/bench_$TAG
mkdir -p $O
cd $R
last() { grep '^{' | tail -1; }
# the two headline lines: full contract (roofline from the replayed trace, CPU baseline)
# (--pmc-table: the per-kernel counters of the run's own PMC child passes -- MFMA busy, observed vs algorithmic bytes)
timeout 900 python bench.py --pmc-table $O/${TAG}_%m_pmc_by_kernel.csv 2> $O/train.err | last > $O/${TAG}_bench_train.json
timeout 900 python bench.py --mode sample --pmc-table $O/${TAG}_%m_pmc_by_kernel.csv 2> $O/sample.err | last > $O/${TAG}_bench_sample.json
# secondary configurations (no CPU baseline leg)
timeout 600 python bench.py --mode sample --dtype fp8 --no-cpu-baseline 2> $O/sample_fp8.err | last > $O/${TAG}_bench_sample_fp8.json
timeout 600 python bench.py --config coco224 --mode sample --dtype bf16 --no-cpu-baseline 2> $O/coco_bf16.err | last > $O/${TAG}_bench_coco_bf16.json
timeout 600 python bench.py --config coco224 --mode sample --dtype fp8 --no-cpu-baseline 2> $O/coco_fp8.err | last > $O/${TAG}_bench_coco_fp8.json
timeout 600 python bench.py --config coco224 --mode train --no-cpu-baseline 2> $O/coco_train.err | last > $O/${TAG}_bench_coco_train.json
# the fp8 train step (forward 3x3 convolutions of the denoiser on e4m3fn operands, device-side weight scales): configs[4] and [1]
timeout 600 python bench.py --config coco224 --mode train --dtype fp8 --no-cpu-baseline --no-pmc 2> $O/coco_train_fp8.err | last > $O/${TAG}_bench_coco_train_fp8.json
timeout 600 python bench.py --mode train --dtype fp8 --no-cpu-baseline --no-pmc 2> $O/train_fp8.err | last > $O/${TAG}_bench_train_fp8.json
# the video configurations (BASELINE configs[2] / [3]) through the same contract
timeout 900 python bench.py --config movid11x6 --no-cpu-baseline --no-pmc 2> $O/movid.err | last > $O/${TAG}_bench_movid11x6.json
timeout 900 python bench.py --config movie15x6 --no-cpu-baseline --no-pmc 2> $O/movie.err | last > $O/${TAG}_bench_movie15x6.json
wc -c $O/*.json
