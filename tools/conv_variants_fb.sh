#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for L in cocosnet_amd/lib/libcocos_hip_cs*.so; do
  echo -n "$(basename $L): "; COCOS_LIB_PATH=$PWD/$L timeout 100 python tools/conv_bench.py 8 2>&1 | tail -1
done
