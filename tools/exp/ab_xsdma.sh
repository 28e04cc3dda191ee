cd $GRAFT_REPO_ROOT
O=gpurun_out/xsdma.txt
python -m pytest tests/test_gpu_igemm_big.py tests/test_gpu_kernels.py -x -q -k "extra" 2>&1 | tail -4 > $O
python -m pytest tests/test_gpu_model.py -x -q -k "dpm or sampl" 2>&1 | tail -3 >> $O
B="python bench.py --steps 8 --warmup 2 --big-batch 0 --no-cpu-baseline --no-roofline --no-pmc"
for i in 1 2 3; do
  for v in 4 3; do
    echo "sample IGEMM_DMA=$v" >> $O
    SDMI_IGEMM_DMA=$v $B --mode sample 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" >> $O
  done
done
for i in 1 2 3; do
  for v in 4 3; do
    echo "train IGEMM_DMA=$v" >> $O
    SDMI_IGEMM_DMA=$v $B --mode train --only-train --steps 30 --warmup 5 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" >> $O
  done
done
