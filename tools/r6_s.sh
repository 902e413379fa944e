#!/bin/bash
# round 6, call S: 8-wave weight-gradient kernel — projection tests, bench-config fp64 tests, step timings + kernel stats
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd); O=$R/gpurun_out/r6_s; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_proj_norm.py tests/test_gpu_baseline_sizes.py tests/test_gpu_plane_prep.py -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -2
for mk in 1 3 1 3; do timeout 300 python tools/step_bench.py --iters 300 --match-kernel $mk 2>&1 | grep match_kernel; done
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -o b -- python $R/tools/step_bench.py --iters 30 --match-kernel 1 > $O/log1.txt 2>&1
python $R/tools/rocprof_summary.py "$(find $O/stats1 -name "*kernel_stats.csv" | head -1)" $O/mk1_kernel_stats.txt > /dev/null 2>&1
head -16 $O/mk1_kernel_stats.txt | cut -c1-130
