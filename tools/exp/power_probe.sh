#!/bin/bash
# Power / clock samples (rocm-smi) while the replayed train step or sampling pass runs: is the step power-limited?
#   MODE=train|sample bash tools/exp/power_probe.sh   -> gpurun_out/power_<mode>.txt
cd $GRAFT_REPO_ROOT
M=${MODE:-train}
if [ "$M" = train ]; then ARGS="--only-train --no-cpu-baseline --no-roofline --steps 600 --warmup 3"; else ARGS="--mode sample --big-batch 0 --no-cpu-baseline --no-roofline --steps 200 --warmup 1"; fi
O=gpurun_out/power_$M.txt
rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -v "^=\|^$" > $O
python bench.py $ARGS > gpurun_out/power_bench_$M.log 2>&1 &
PID=$!
sleep 25        # model build + capture
for i in $(seq 1 20); do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -i "power\|sclk\|mclk\|fclk\|junction\|edge" | tr -s ' ' | tr '\n' '|' >> $O
  echo >> $O
  sleep 0.5
done
wait $PID
tail -1 gpurun_out/power_bench_$M.log | cut -c1-160 >> $O
