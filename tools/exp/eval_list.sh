# launch list of one sampler evaluation (replayed graph) on the current build -> gpurun_out/eval_list.txt
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/evl
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/sample -o sample --output-format csv -- python $R/bench.py --mode sample --big-batch 0 --no-cpu-baseline --no-roofline --steps 3 --warmup 1 > $O/sample_trace.log 2>&1
cd $R
python tools/trace_eval.py $O/sample/sample_kernel_trace.csv > gpurun_out/eval_list.txt 2>&1
rm -rf $O
