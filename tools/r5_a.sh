#!/bin/bash
# round 5, call A: the whole GPU suite + the printed fp64 comparisons (unforced and on the fp64 branch pattern)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5_a; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_gpu.log
timeout 900 python -m pytest tests/test_gpu_conv.py -q -s -m gpu -k "end_to_end or config3" 2>&1 | grep "E2E_FP64\|CFG3_FP64\|passed\|failed\|Error\|assert" > $O/e2e_fp64_errors.txt
cat $O/e2e_fp64_errors.txt | cut -c1-3000
