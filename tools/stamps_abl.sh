#!/bin/bash
# on the GPU box: in-kernel stamps of chosen conv_ws launches with the diagnostic and the ablation builds (tools/build_dbg.sh, ABL="1 3")
#   tools/stamps_abl.sh <tag> "<shape> <shape> ..."   -> gpurun_out/<tag>_stamps.txt
tag=$1
out=gpurun_out/${tag}_stamps.txt
: > $out
for lib in dbg $(cd unet_amd && ls libsolpex_hip_dbg_abl*.so 2>/dev/null | sed 's/libsolpex_hip_//; s/.so//'); do
  cp unet_amd/libsolpex_hip_$lib.so unet_amd/libsolpex_hip.so
  for shape in $2; do
    echo "== SHAPE=$shape build=$lib" >> $out
    SHAPE=$shape timeout -k 10 60 python tools/conv_stamps.py ${STAMP_B:-64} >> $out 2>&1 || exit 1
  done
done
