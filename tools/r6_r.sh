#!/bin/bash
# round 6, call R: lazy-rescale threshold of the K2 forward (thr + bias = 15): timing and tail precision per threshold
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
bash tools/ab_libs.sh libcocos_hip.so libcocos_thr8.so libcocos_thr9.so libcocos_thr10.so 2>&1 | cut -c1-200
for L in libcocos_hip.so libcocos_thr8.so libcocos_thr9.so libcocos_thr10.so; do
  echo "== $L"; COCOS_LIB_PATH=$PWD/cocosnet_amd/lib/$L timeout 600 python -m pytest tests/test_gpu_mk3_sizes.py tests/test_gpu_parity.py -q -m gpu -k "elementwise or values_far or peaked" 2>&1 | tail -4 | cut -c1-200
  COCOS_LIB_PATH=$PWD/cocosnet_amd/lib/$L timeout 300 python tools/precision_check.py 2>&1 | tail -4 | cut -c1-300
done
