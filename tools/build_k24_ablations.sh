#!/bin/bash
# Debug libraries that differ from the product build in proj_bwd_f16x3.hip only.  usage: tools/build_k24_ablations.sh "1 2 4 8"
cd "$(dirname "$0")/.."
L=cocosnet_amd/lib; F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-function"
OBJS=$(ls $L/obj/*.o | grep -v "proj_bwd_f16x3")
for a in ${1:-0}; do
  (hipcc $F -DCOCOS_K24_ABLATE=$a -c cocosnet_amd/csrc/proj_bwd_f16x3.hip -o /tmp/k24_$a.o &&
   hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libcocos_hip_k24abl$a.so $OBJS /tmp/k24_$a.o && echo built $a) &
done
wait
