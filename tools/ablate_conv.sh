#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
echo -n "product: "; timeout 100 python tools/conv_fwd_ms.py "$@"
for L in cocosnet_amd/lib/libcocos_hip_cabl*.so; do
  echo -n "$(basename $L): "; COCOS_LIB_PATH=$PWD/$L timeout 100 python tools/conv_fwd_ms.py "$@"
done
