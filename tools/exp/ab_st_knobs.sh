# same-box A/B of two module-level constants of the fused training path (no environment switch exists for them)
cd $GRAFT_REPO_ROOT
run() {
python - "$@" <<'PY'
import sys, runpy
import slotdiffusion_amd.kern as k
wq, minwgs = float(sys.argv[1]), int(sys.argv[2])
k.WeightBank.WQ_TARGET = wq
k._ST_TRAIN_MIN_WGS = minwgs
sys.argv = ['bench.py', '--steps', '20', '--warmup', '3', '--only-train', '--no-cpu-baseline', '--no-roofline', '--no-pmc']
runpy.run_path('bench.py', run_name='__main__')
PY
}
rm -f gpurun_out/ab_st_knobs.txt
for i in 1 2; do
  for v in "512 96" "256 96" "768 96" "384 96" "512 200"; do
    echo "WQ_TARGET MIN_WGS = $v" >> gpurun_out/ab_st_knobs.txt
    run $v 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" >> gpurun_out/ab_st_knobs.txt
  done
done
python -m pytest tests/test_gpu_st_train.py -x -q -k torch_autograd 2>&1 | tail -8 >> gpurun_out/ab_st_knobs.txt
