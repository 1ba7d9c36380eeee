#!/bin/bash
# Diagnostic library with in-kernel s_memtime stamps in conv_ws_kernel: unet_amd/libsolpex_hip_dbg.so (git-ignored).
# On the GPU box: cp unet_amd/libsolpex_hip_dbg.so unet_amd/libsolpex_hip.so (the box's copy of the tree is scratch), then tools/conv_stamps.py
# ABL=1|2|3 adds ablation builds libsolpex_hip_dbg_abl<N>.so (SX_WS_ABL: wrong results, timing only)
set -e
cd "$(dirname "$0")/../unet_amd/csrc"
objs=$(ls build/*.o | grep -v conv_ws.o | grep -v _dbg)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DSX_WS_DBG=1 -c conv_ws.hip -o build/conv_ws_dbg.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs build/conv_ws_dbg.o -o ../libsolpex_hip_dbg.so
for n in $ABL; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DSX_WS_DBG=1 -DSX_WS_ABL=$n -c conv_ws.hip -o build/conv_ws_dbg_abl$n.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs build/conv_ws_dbg_abl$n.o -o ../libsolpex_hip_dbg_abl$n.so
done
