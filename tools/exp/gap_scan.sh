R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/gaps
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $O/t -o t --output-format csv -- python $R/bench.py --only-train --no-cpu-baseline --no-roofline --no-pmc --steps 8 --warmup 3 > $O/log.txt 2>&1
cd $R
python tools/exp/gap_scan.py $O/t/t_kernel_trace.csv > gpurun_out/gap_scan.txt 2>&1
rm -rf $O
