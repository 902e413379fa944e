#!/bin/bash
# Debug: the x-box correlation GEMM (hgemm_f16x3.hip, EPI 1 / 2) under ablations.  usage: tools/xbox_ablate.sh build "1 3 4 8" | run "0 1 3 4 8"
cd "$(dirname "$0")/.."
L=cocosnet_amd/lib
if [ "$1" = build ]; then
  OBJS=$(ls $L/obj/*.o | grep -v "hgemm_f16x3")
  for a in $2; do
    (hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-function -DHG_ABLATE=$a -c cocosnet_amd/csrc/hgemm_f16x3.hip -o /tmp/hg_abl_$a.o &&
     hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libcocos_hip_hg$a.so $OBJS /tmp/hg_abl_$a.o && echo built $a) &
  done; wait
else
  for a in $2; do
    Lp=$PWD/$L/libcocos_hip_hg$a.so; [ "$a" = 0 ] && Lp=$PWD/$L/libcocos_hip.so
    echo -n "ablate $a: "; COCOS_LIB_PATH=$Lp timeout 120 python tools/xbox_bench.py 2>&1 | tail -1
  done
fi
