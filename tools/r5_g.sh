#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5_g; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_baseline_sizes.py -q -m gpu -k "contextual" 2>&1 | tail -2
timeout 600 python tools/contextual_bench.py 2>/dev/null | grep "^{" > $O/contextual_bench.txt; cat $O/contextual_bench.txt | cut -c1-330
