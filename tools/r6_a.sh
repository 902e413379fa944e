#!/bin/bash
# round 6, call A: MFMA ceiling with clocks, the GPU suite on the untouched kernels (+ the routing / hygiene edits), a short bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r6_a; rm -rf $O; mkdir -p $O
bash tools/mfma_ceiling.sh > $O/mfma_ceiling.log 2>&1; tail -40 $O/mfma_ceiling.log
timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest_gpu.log
timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 1500 $O/bench.json
