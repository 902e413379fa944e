#!/bin/bash
# forward time of K16 under alternate builds cocosnet_amd/lib/libcocos_hip_cs*.so
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for bn in 128 256; do export COCOS_CONV_BN=$bn
for L in cocosnet_amd/lib/libcocos_hip_cs*.so; do
  echo -n "BN=$bn $(basename $L): "; COCOS_LIB_PATH=$PWD/$L timeout 100 python tools/conv_fwd_ms.py "$@"
done; done
