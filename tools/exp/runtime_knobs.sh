#!/bin/bash
# ROCm runtime knobs vs the graphed train step and sampler (ms per step / per 20-NFE pass), one box
P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],2))'
run() {
  echo -n "$1 => train "; env $1 python bench.py --only-train --no-cpu-baseline --no-roofline --steps 20 --warmup 5 2>&1 | tail -1 | python -c "$P" 2>&1 | tail -1
  echo -n "$1 => sample "; env $1 python bench.py --mode sample --no-cpu-baseline --no-roofline --big-batch 0 --steps 4 --warmup 2 2>&1 | tail -1 | python -c "$P" 2>&1 | tail -1
}
run "SDMI_NOP=1"
run "HIP_FORCE_DEV_KERNARG=1"
run "HIP_FORCE_DEV_KERNARG=0"
run "GPU_MAX_HW_QUEUES=2"
run "GPU_MAX_HW_QUEUES=8"
run "AMD_DIRECT_DISPATCH=0"
run "HSA_ENABLE_SDMA=0"
run "SDMI_NOP=2"
