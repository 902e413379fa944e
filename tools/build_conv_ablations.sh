#!/bin/bash
# Debug libraries with -DCOCOS_CONV_ABLATE=<bits> (conv_f16x3.hip only differs).  usage: tools/build_conv_ablations.sh "0 1 2 4 8"
cd "$(dirname "$0")/.."
L=cocosnet_amd/lib; F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-function -Iinclude $COCOS_ABL_EXTRA"
OBJS=$(ls $L/obj/*.o | grep -v "conv_f16x3")
for a in ${1:-0}; do
  (hipcc $F -DCOCOS_CONV_ABLATE=$a -c cocosnet_amd/csrc/conv_f16x3.hip -o /tmp/abl_conv_$a.o &&
   hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libcocos_hip_cabl$a.so $OBJS /tmp/abl_conv_$a.o && echo built $a) &
done
wait
