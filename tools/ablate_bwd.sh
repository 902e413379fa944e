#!/bin/bash
# Debug: kernel times of the split kernels under ablations (libraries from tools/build_ablations.sh); wrong results, real timing
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for a in 0 1 2 4 8 16 3 12 31; do
  L=$PWD/cocosnet_amd/lib/libcocos_hip_abl$a.so
  [ -f $L ] || continue
  echo "== ablate $a (1 staging, 2 operand reads, 4 exp, 8 dS'' stores, 16 logits loads)"
  COCOS_LIB_PATH=$L timeout 120 python tools/kernel_bench.py --iters 10 2>&1 | grep -E "^train" 
  COCOS_LIB_PATH=$L timeout 120 python tools/phase_timing_f16x3.py 154 train 2>&1 | grep -E "^bwd" | tail -1
done
