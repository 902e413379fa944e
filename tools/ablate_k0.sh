#!/bin/bash
# kernel times of the K0 streaming kernels for the product library and every libcocos_hip_k0abl*.so (tools/build_k0_ablations.sh)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
echo "== product"; tools/k0_prof.sh "$@" | grep proj_stream
for L in cocosnet_amd/lib/libcocos_hip_k0abl*.so; do
  echo "== $(basename $L)"; COCOS_LIB_PATH=$PWD/$L tools/k0_prof.sh "$@" | grep proj_stream
done
