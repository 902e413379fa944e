#!/bin/bash
# K13 / K9 with and without the max|.| by-product (COCOS_CONV=bf16 selects the plain entry points), same box; then the module scope
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 600 python -m pytest tests/test_gpu_amax_epilogues.py -q -m gpu -x 2>&1 | tail -3
echo "== K13 with amax"; python tools/instnorm_bench.py 2>&1 | grep shape | cut -c1-200
echo "== K13 plain";     COCOS_CONV=bf16 python tools/instnorm_bench.py 2>&1 | grep shape | cut -c1-200
echo "== K9 with amax";  python tools/pono_bench.py 8 512 64 64 2>&1 | tail -2 | cut -c1-300
echo "== K9 plain";      COCOS_CONV=bf16 python tools/pono_bench.py 8 512 64 64 2>&1 | tail -2 | cut -c1-300
bash tools/ab_old_new_netcorr.sh
