cd $GRAFT_REPO_ROOT
B="python bench.py --steps 20 --warmup 3 --only-train --no-cpu-baseline --no-roofline --no-pmc"
rm -f gpurun_out/ab_st_train.txt
for i in 1 2 3; do
  for v in 0 1 2; do
    echo "ATTN_BWD_SLOTS=$v" >> gpurun_out/ab_st_train.txt
    SDMI_ATTN_BWD_SLOTS=$v $B 2>gpurun_out/ab_err.txt | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d.get('loss'))" >> gpurun_out/ab_st_train.txt
  done
done
