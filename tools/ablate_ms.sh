#!/bin/bash
# kernel times of the K2 split kernels under ablations (UNinstrumented libraries from tools/build_ablations.sh)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for rep in 1; do
for a in 0 256 0 256; do
  L=$PWD/cocosnet_amd/lib/libcocos_hip_abl$a.so
  [ -f $L ] || continue
  echo -n "abl $a: "
  COCOS_LIB_PATH=$L timeout 120 python tools/kernel_bench.py --iters 20 2>&1 | grep -E "^train +corr" | awk '{printf "%s %s  ", $2, $6}'; echo
done; done
