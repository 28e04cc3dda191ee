#!/bin/bash
# us per launch of the 32^2 / 64^2 convolution chains for each ablation variant (results of a variant are wrong by design)
cd $GRAFT_REPO_ROOT
for m in "$@"; do
  L=tools/exp/libsdmi_pp_x$m.so; [ "$m" = base ] && L=slotdiffusion_amd/libsdmi.so
  SDMI_LIBPATH=$L timeout 300 python tools/exp/conv_chain.py 2>&1 | grep "conv3x3" | grep -E "H= 32|H= 64" | sed "s/^DMA=3 LW=8 ALL=0/x$m/"
done
