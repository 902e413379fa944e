#!/bin/bash
# final check of the round: full GPU suite, smoke, contextual artefacts, default bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5_h; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/r05_pytest_gpu_final.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r05_pytest_gpu_final.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 300 python -m pytest tests/test_gpu_baseline_sizes.py -q -s -m gpu -k "contextual" 2>&1 | grep "CTX_FP64\|passed\|failed" | cut -c1-600 > $O/r05_contextual_fp64_errors.txt
timeout 300 python tools/contextual_bench.py 2>/dev/null | grep "^{" > $O/r05_contextual_bench.txt
timeout 900 python bench.py > $O/r05_bench_final.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-300 $O/r05_bench_final.json
