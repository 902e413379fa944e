#!/bin/bash
# Debug libraries for tools/ablate_*.sh: only the two K2 split translation units differ, the rest is reused from the
# product build (objects under cocosnet_amd/lib/obj).   usage: tools/build_ablations.sh "0 1 2 4 8 16 31"
cd "$(dirname "$0")/.."
L=cocosnet_amd/lib; F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-function $COCOS_ABL_EXTRA"
OBJS=$(ls $L/obj/*.o | grep -v "corr_fused_fwd_f16x3\|corr_fused_bwd_f16x3")
for a in ${1:-0}; do
  (hipcc $F -DCOCOS_ABLATE=$a -c cocosnet_amd/csrc/corr_fused_fwd_f16x3.hip -o /tmp/abl_fwd_$a.o &
   hipcc $F -DCOCOS_ABLATE=$a -c cocosnet_amd/csrc/corr_fused_bwd_f16x3.hip -o /tmp/abl_bwd_$a.o & wait
   hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libcocos_hip_abl$a.so $OBJS /tmp/abl_fwd_$a.o /tmp/abl_bwd_$a.o && echo built $a) &
done
wait
