#!/bin/bash
# round 6, call I: warp head (one backward kernel for d out + max + D), pair split — tests, parity, step timing
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd); O=$R/gpurun_out/r6_i; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_plane_prep.py tests/test_gpu_proj_norm.py tests/test_gpu_parity.py -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
for mk in 1 3; do timeout 300 python tools/step_bench.py --iters 300 --match-kernel $mk 2>&1 | grep match_kernel; done
export TMPDIR=/tmp; cd /tmp
for mk in 1 3; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats$mk -o b -- python $R/tools/step_bench.py --iters 30 --match-kernel $mk > $O/log$mk.txt 2>&1
python $R/tools/rocprof_summary.py "$(find $O/stats$mk -name "*kernel_stats.csv" | head -1)" $O/mk${mk}_kernel_stats.txt > /dev/null 2>&1
head -30 $O/mk${mk}_kernel_stats.txt | cut -c1-140
done
