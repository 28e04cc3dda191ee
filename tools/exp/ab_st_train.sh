cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_st_train.py -x -q 2>&1 | tail -6 > gpurun_out/st_train_t10.txt
python tools/exp/st_repeat_model.py 20 2>&1 | grep -v amdgpu >> gpurun_out/st_train_t10.txt
B="python bench.py --steps 20 --warmup 3 --only-train --no-cpu-baseline --no-roofline --no-pmc"
rm -f gpurun_out/ab_st_train.txt
for i in 1 2 3; do
  for v in 0 1; do
    echo "ST_BWD_MERGED=$v" >> gpurun_out/ab_st_train.txt
    SDMI_ST_BWD_MERGED=$v $B 2>gpurun_out/ab_err.txt | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d.get('loss'))" >> gpurun_out/ab_st_train.txt
  done
done
bash tools/exp/trace_train.sh merged
