#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5_e; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_baseline_sizes.py -q -s -m gpu -k "contextual" > $O/ctx.log 2>&1; echo "ctx rc=$?"
grep "CTX_FP64\|passed\|failed\|Error\|error" $O/ctx.log | cut -c1-400 | head -20
timeout 600 python tools/contextual_bench.py > $O/contextual_bench.txt 2>&1; cat $O/contextual_bench.txt | cut -c1-400
timeout 900 python -m pytest tests/test_gpu_mk3_sizes.py -q -s -m gpu -k "config5" 2>&1 | grep "CFG5_MK3\|passed\|failed" | cut -c1-300
