#!/bin/bash
# Debug: phase timing of the split forward under ablations (libraries built by tools/build_ablations.sh).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for a in 0 1 2 4 8 3 7 15; do
  L=$PWD/cocosnet_amd/lib/libcocos_hip_abl$a.so
  [ -f $L ] || continue
  echo "== ablate $a (1 staging, 2 operand reads, 4 softmax, 8 logits store)"
  COCOS_LIB_PATH=$L timeout 120 python tools/phase_timing_f16x3.py 154 train 2>&1 | grep -E "^QK" | tail -1
done
