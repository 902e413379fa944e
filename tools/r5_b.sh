#!/bin/bash
# round 5, call B: K22 (contextual loss without [N,N]) + the whole GPU suite
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5_b; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_baseline_sizes.py -q -s -m gpu -k "contextual" > $O/ctx.log 2>&1; echo "ctx rc=$?"
grep "CTX_FP64\|passed\|failed\|Error\|error" $O/ctx.log | cut -c1-600 | head -40
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest_gpu.log | cut -c1-400
