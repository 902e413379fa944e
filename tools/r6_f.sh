#!/bin/bash
# round 6, call F: K24 tests, then the step A/B (backward fused on / off) for match_kernel 1 and 3
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd); O=$R/gpurun_out/r6_f; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_proj_norm.py -q -m gpu -x -s > $O/pytest_k24.log 2>&1; echo "k24 rc=$?"; grep -E "K24_|passed|failed|Error|assert" $O/pytest_k24.log | cut -c1-260 | tail -50
for mk in 1 3; do for f in 0 1 0 1; do
  COCOS_PROJ_BWD_FUSED=$f timeout 300 python tools/step_bench.py --iters 300 --match-kernel $mk 2>&1 | grep match_kernel | sed "s/^/bwd_fused=$f /"
done; done
