# kernel trace of the replayed train step -> breakdown + ordered launch list (gpurun_out/trace_<tag>_*.txt)
TAG=${1:-cur}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/trace_tmp
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $O -o t --output-format csv -- python $R/bench.py --only-train --no-cpu-baseline --no-roofline --no-pmc --steps 4 --warmup 2 > $O/log.txt 2>&1
cd $R
python tools/trace_step.py $O/t_kernel_trace.csv 70 > gpurun_out/trace_${TAG}_breakdown.txt 2>&1
python tools/trace_step_list.py $O/t_kernel_trace.csv > gpurun_out/trace_${TAG}_list.txt 2>&1
rm -rf $O
