#!/bin/bash
# Same-box A/B of the symmetric-wave igemm kernel (igemm_sym.h): dependent conv chains + correctness, then the
# kernel unit tests with the kernel forced on.  usage: bash tools/exp/sym_ab.sh [stages...]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for s in 0 ${@:-4 3 2}; do
  echo "=== SDMI_IGEMM_SYM=$s" | tee -a gpurun_out/sym_ab.txt
  SDMI_IGEMM_SYM=$s timeout 600 python tools/exp/conv_chain.py 2>&1 | grep -v Warning | tee -a gpurun_out/sym_ab.txt
done
SDMI_IGEMM_SYM=${1:-4} timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "igemm or conv or linear or gemm" 2>&1 | tail -5 | tee -a gpurun_out/sym_ab.txt
