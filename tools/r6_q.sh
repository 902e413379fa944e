#!/bin/bash
# round 6, call Q: single accumulator chains (K2 forward QK, K19' dP) — parity + baseline-size tests, step timings
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd); O=$R/gpurun_out/r6_q; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_sizes.py tests/test_gpu_mk3_sizes.py -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for mk in 1 3 1 3; do timeout 300 python tools/step_bench.py --iters 300 --match-kernel $mk 2>&1 | grep match_kernel; done
