#!/bin/bash
# Build a variant of libsdmi.so with extra -D flags for ONE source (kernel studies; not part of the product):
#   bash tools/exp/build_variant.sh igemm.hip pp_tl -DSDMI_PP_TIMELINE      -> tools/exp/libsdmi_pp_tl.so
# The other objects come from the product build (slotdiffusion_amd/csrc/_build).
set -e
cd "$(dirname "$0")/../.."
SRC=$1; NAME=$2; shift 2
python -c "from slotdiffusion_amd.csrc.build import build; build()"
B=slotdiffusion_amd/csrc/_build
OBJ=tools/exp/_var/${NAME}_${SRC%.*}.o
EXTRA=""
case $SRC in vq.hip|elementwise.hip) EXTRA="-ffp-contract=off";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $EXTRA "$@" -x hip -c slotdiffusion_amd/csrc/$SRC -o $OBJ
OBJS=$(ls $B/*.o | grep -v "/${SRC%.*}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/libsdmi_${NAME}.so $OBJS $OBJ
echo tools/exp/libsdmi_${NAME}.so
