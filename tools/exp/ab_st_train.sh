cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_st_train.py -x -q 2>&1 | tail -15 > gpurun_out/st_train_t2.txt
python -m pytest tests/test_gpu_train.py -x -q -k "one_split" 2>&1 | tail -3 >> gpurun_out/st_train_t2.txt
B="python bench.py --steps 20 --warmup 3 --only-train --no-cpu-baseline --no-roofline --no-pmc"
for i in 1 2; do
  for v in 0 1; do
    echo "ST_TRAIN=$v" >> gpurun_out/ab_st_train.txt
    SDMI_ST_TRAIN=$v $B 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d.get('loss'))" >> gpurun_out/ab_st_train.txt
  done
done
