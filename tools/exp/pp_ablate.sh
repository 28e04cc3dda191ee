#!/bin/bash
# Build variants of igemm_pp.hip (igemm_pp.h: SDMI_PP_EXP ablation mask, SDMI_PP_KSEG), un-instrumented: wall time per
# launch through tools/exp/conv_chain.py is the measurement.   bash tools/exp/pp_ablate.sh 0 1 k4:0 k4:2 ...   (TL=1: timeline)
cd "$(dirname "$0")/../.."
for v in "$@"; do
  m=${v#*:}; k=2; [[ $v == k4:* ]] && k=4
  bash tools/exp/build_variant.sh igemm_pp.hip pp_x${v/:/_} ${TL:+-DSDMI_PP_TIMELINE} -DSDMI_PP_EXP=$m -DSDMI_PP_KSEG=$k 2>&1 | tail -1
done
