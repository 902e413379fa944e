#!/bin/bash
# Debug libraries with -DPS_ABLATE=<bits> (proj_stream_f16x3.hip only differs).  usage: tools/build_k0_ablations.sh "1 2 4 8" [extra flags]
cd "$(dirname "$0")/.."
L=cocosnet_amd/lib; F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-function -Iinclude $2"
OBJS=$(ls $L/obj/*.o | grep -v "proj_stream_f16x3")
rm -f $L/libcocos_hip_k0abl*.so
for a in ${1:-0}; do
  (hipcc $F -DPS_ABLATE=$a -c cocosnet_amd/csrc/proj_stream_f16x3.hip -o /tmp/abl_ps_$a.o &&
   hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libcocos_hip_k0abl$a.so $OBJS /tmp/abl_ps_$a.o && echo built $a) &
done
wait
