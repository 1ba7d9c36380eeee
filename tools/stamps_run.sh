#!/bin/bash
# on the GPU box: in-kernel stamps of the deep-level conv_ws launches, K split on and off -> gpurun_out/<tag>_stamps.txt
tag=$1
cp unet_amd/libsolpex_hip_dbg.so unet_amd/libsolpex_hip.so
out=gpurun_out/${tag}_stamps.txt
: > $out
for shape in c3,16,8,256,256 c3,32,16,128,128 dual,32,16,256,128 dual,16,8,128,256 dual,64,32,128,64 center,16,8,256,128 ${EXTRA_SHAPES}; do
  for ks in 1 0; do
    echo "== SHAPE=$shape SX_CONV_KQ=$ks" >> $out
    SHAPE=$shape SX_CONV_KQ=$ks timeout -k 10 60 python tools/conv_stamps.py ${STAMP_B:-64} >> $out 2>&1 || exit 1
  done
done
