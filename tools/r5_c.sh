#!/bin/bash
# round 5, call C: K22 with 64-row blocks, its bench against the materialised route, the conv flavour table (emulated arms)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5_c; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_baseline_sizes.py -q -s -m gpu -k "contextual" > $O/ctx.log 2>&1; echo "ctx rc=$?"
grep "CTX_FP64\|passed\|failed\|Error\|error" $O/ctx.log | cut -c1-400 | head -20
timeout 600 python tools/contextual_bench.py > $O/contextual_bench.txt 2>&1; cat $O/contextual_bench.txt | cut -c1-400
timeout 900 python tools/conv_flavour_table.py $O/conv_flavour_table.json > $O/conv_flavour_table.txt 2>&1; cut -c1-700 $O/conv_flavour_table.txt | tail -12
