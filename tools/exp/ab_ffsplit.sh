cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_st_fused.py -x -q 2>&1 | tail -8 > gpurun_out/ffsplit.txt
python -m pytest tests/test_gpu_model.py -x -q -k "dpm or sampl" 2>&1 | tail -3 >> gpurun_out/ffsplit.txt
B="python bench.py --mode sample --steps 8 --warmup 2 --big-batch 0 --no-cpu-baseline --no-roofline --no-pmc"
for i in 1 2 3; do
  for v in 0 1; do
    echo "ST_FF_SPLIT=$v" >> gpurun_out/ffsplit.txt
    SDMI_ST_FF_SPLIT=$v $B 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" >> gpurun_out/ffsplit.txt
  done
done
for c in ; do
  for v in 0 1; do
    echo "$c ST_FF_SPLIT=$v" >> gpurun_out/ffsplit.txt
    SDMI_ST_FF_SPLIT=$v $B --config $c 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" >> gpurun_out/ffsplit.txt
  done
done
