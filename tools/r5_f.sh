#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5_f; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_baseline_sizes.py -q -s -m gpu -k "contextual" > $O/ctx.log 2>&1; echo "ctx rc=$?"
grep "CTX_FP64_\|passed\|failed\|Error\|error" $O/ctx.log | cut -c1-500 | head -20
timeout 600 python tools/contextual_bench.py 2>/dev/null | grep "^{" > $O/contextual_bench.txt; cat $O/contextual_bench.txt | cut -c1-400
