#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for bn in 128 256; do
export COCOS_CONV_BN=$bn
echo -n "BN=$bn product: "; timeout 100 python tools/conv_fwd_ms.py "$@"
for a in 1 2 4 7; do
  L=$PWD/cocosnet_amd/lib/libcocos_hip_cabl$a.so
  [ -f $L ] || continue
  echo -n "BN=$bn abl $a: "; COCOS_LIB_PATH=$L timeout 100 python tools/conv_fwd_ms.py "$@"
done; done
