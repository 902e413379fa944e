#!/bin/bash
# round 5, call D: cfg5 + mk3 peak memory with the dC planes in T's storage; full suite
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5_d; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mk3_sizes.py -q -s -m gpu -k "config5" 2>&1 | grep "CFG5_MK3\|passed\|failed\|Error\|assert" | cut -c1-300
COCOS_BOX3_ALIAS_T_BYTES=99999999999999 timeout 900 python -m pytest tests/test_gpu_mk3_sizes.py -q -s -m gpu -k "config5" 2>&1 | grep "CFG5_MK3\|passed\|failed" | cut -c1-300
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log | cut -c1-300
