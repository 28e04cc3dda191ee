R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/evaltrace
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/sample -o sample --output-format csv -- python $R/bench.py --mode sample --big-batch 0 --no-cpu-baseline --no-roofline --steps 2 --warmup 1 > $O/sample_trace.log 2>&1
cd $R
grep -E "vq_kernel|gn_fused2|splitk" $O/sample/sample_kernel_stats.csv | cut -c1-200
python tools/trace_eval.py $O/sample/sample_kernel_trace.csv > $O/eval_list.txt 2>&1
rm -rf $O/sample
head -1 $O/eval_list.txt
