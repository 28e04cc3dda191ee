#!/bin/bash
# 8^2 / 4^2-level 3x3 convolutions: LDS-DMA kernel on 64 x 64 tiles (tools/exp/conv_chain.py SHAPES=low)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export SHAPES=low
for t in 0 2048; do
  SDMI_IGEMM_DMA64=$t timeout 300 python tools/exp/conv_chain.py | sed "s/^/DMA64=$t /"
done 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/lowres_sweep.txt
