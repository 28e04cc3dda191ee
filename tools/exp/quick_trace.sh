R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/q1; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/train -o train --output-format csv -- python $R/bench.py --only-train --no-cpu-baseline --no-roofline --steps 5 --warmup 2 > $O/train_trace.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $O/sample -o sample --output-format csv -- python $R/bench.py --mode sample --big-batch 0 --no-cpu-baseline --no-roofline --steps 3 --warmup 1 > $O/sample_trace.log 2>&1
cd $R
python tools/trace_step.py $O/train/train_kernel_trace.csv 70 > $O/train_step_breakdown.txt 2>&1
cp $O/sample/sample_kernel_stats.csv $O/sample_stats.csv; cp $O/train/train_kernel_stats.csv $O/train_stats.csv
rm -rf $O/train $O/sample
head -50 $O/train_step_breakdown.txt
