#!/bin/bash
# Ablation libraries of the fused match_kernel-3 kernels (box3_fused_f16x3.hip, -DBX_ABLATE=<bits>, results WRONG, timing
# real) and their kernel times at the cfg2' shape.   build (CPU box): tools/box3_ablate.sh build "0 1 2 4 8 16 32"
#                                                    time  (GPU box): tools/box3_ablate.sh run   "0 1 2 4 8 16 32"
cd "$(dirname "$0")/.."
L=cocosnet_amd/lib; F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-function"
if [ "$1" = build ]; then
  OBJS=$(ls $L/obj/*.o | grep -v "box3_fused_f16x3")
  for a in ${2:-0}; do
    (hipcc $F -DBX_ABLATE=$a -c cocosnet_amd/csrc/box3_fused_f16x3.hip -o /tmp/bx_abl_$a.o &&
     hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libcocos_hip_bx$a.so $OBJS /tmp/bx_abl_$a.o && echo built $a) &
  done
  wait
else
  for a in ${2:-0}; do
    LIB=$PWD/$L/libcocos_hip_bx$a.so
    [ -f $LIB ] || continue
    echo -n "BX_ABLATE=$a: "
    COCOS_LIB_PATH=$LIB timeout 120 python tools/box3_bench.py 2>&1 | tail -1
  done
fi
