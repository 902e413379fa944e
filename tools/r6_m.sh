#!/bin/bash
# round 6, call M: the whole GPU suite after the hygiene refactor (tape out of the package, oracle arbiter for K22, fp64 PReLU gradient)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd); O=$R/gpurun_out/r6_m; rm -rf $O; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu -x -s > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
grep -E "E2E_FLIPS|prelu" $O/pytest_gpu.log | cut -c1-300 | head -20
