#!/bin/bash
# on the GPU box: conv_kq stage-time ablations (diagnostic libraries libsolpex_hip_abl{1,2}.so: no MFMAs / no fragment reads)
tag=$1
out=gpurun_out/${tag}_abl.txt
: > $out
for lib in dbg abl1 abl2; do
  cp unet_amd/libsolpex_hip_$lib.so unet_amd/libsolpex_hip.so
  for shape in c3,16,8,256,256 c3,32,16,128,128 dual,16,8,128,256; do
    echo "== lib=$lib SHAPE=$shape" >> $out
    SHAPE=$shape timeout -k 10 60 python tools/conv_stamps.py 64 >> $out 2>&1 || exit 1
  done
done
