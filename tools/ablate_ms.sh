#!/bin/bash
# kernel times of the K2 split kernels under ablations (UNinstrumented libraries from tools/build_ablations.sh):
#   tools/ablate_ms.sh "0 1 2 4 8 16 24 256"      results are WRONG under ablation, only the timing is real
# bits: 1 no tile staging | 2 no operand re-reads from LDS | 4 no exp | 8 no dS''/logits stores | 16 no saved-logits loads
#       32 stores to one hot block | 64 staging loads from one hot tile | 256 default cache policy instead of nt
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for a in ${1:-0}; do
  L=$PWD/cocosnet_amd/lib/libcocos_hip_abl$a.so
  [ -f $L ] || continue
  echo -n "ablate $a: "
  COCOS_LIB_PATH=$L timeout 120 python tools/kernel_bench.py --iters 20 2>&1 | grep -E "^train +corr" | awk '{printf "%s %s ms   ", $2, $6}'; echo
done
