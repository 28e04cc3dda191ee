#!/bin/bash
# Same-box A/B of the 256 x 128 ping-pong igemm kernel (igemm_pp.h): its parity tests, dependent conv chains with it
# off / on, then the kernel unit tests with it forced on for every eligible shape.  usage: bash tools/exp/pp_ab.sh
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/pp_ab.txt
: > $O
timeout 900 python -m pytest tests/test_gpu_igemm_pp.py -x -q -m gpu 2>&1 | tail -15 | tee -a $O
for s in 0 192; do
  echo "=== SDMI_IGEMM_PP=$s" | tee -a $O
  SDMI_IGEMM_PP=$s timeout 600 python tools/exp/conv_chain.py 2>&1 | grep -v Warning | tee -a $O
done
echo "=== kernel tests, SDMI_IGEMM_PP=1 SDMI_IGEMM_PP_MINKT=1 (every eligible shape)" | tee -a $O
SDMI_IGEMM_PP=1 SDMI_IGEMM_PP_MINKT=1 timeout 1200 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "igemm or conv or linear or gemm" 2>&1 | tail -5 | tee -a $O
