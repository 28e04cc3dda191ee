#!/bin/bash
# A/B of igemm variants on the big convolution shapes (GPU box): SDMI_IGEMM_DMA=0/1
for v in 0 1; do
  for s in "64 32 256 256 3" "64 32 128 128 3" "64 64 128 128 3" "64 32 256 128 3" "64 32 384 128 3" "64 16 256 256 3" "64 16 512 256 3" "64 32 256 128 1" "64 16 1024 256 1" "64 16 256 2048 1" "64 8 384 384 3"; do
    TAG="dma=$v" SDMI_IGEMM_DMA=$v python tools/time_one.py $s 2>/dev/null | tail -1
  done
done
