cd $GRAFT_REPO_ROOT
O=gpurun_out/dist_selftest.txt
echo "== torchrun nproc 1" > $O
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline --no-pmc --no-roofline 2>&1 | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['n_gpus'], d['comm'])" >> $O 2>&1
echo "== forced RCCL path, one rank" >> $O
SDMI_BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pmc --no-roofline 2>&1 | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['n_gpus'], d['comm'])" >> $O 2>&1
echo "== forced RCCL path, sampling" >> $O
SDMI_BENCH_FORCE_DIST=1 timeout 600 python bench.py --mode sample --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-roofline 2>&1 | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['n_gpus'], d['comm'])" >> $O 2>&1
python -m pytest tests/test_gpu_ddp.py -x -q 2>&1 | tail -2 >> $O
