#!/bin/bash
# round 6, call O: box3 kernels with a compiler-visible uniform wave index (no waterfall loops) — mk3 tests + kernel stats
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd); O=$R/gpurun_out/r6_o; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mk3_sizes.py tests/test_gpu_parity.py -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for mk in 3 3; do timeout 300 python tools/step_bench.py --iters 300 --match-kernel $mk 2>&1 | grep match_kernel; done
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats3 -o b -- python $R/tools/step_bench.py --iters 30 --match-kernel 3 > $O/log3.txt 2>&1
python $R/tools/rocprof_summary.py "$(find $O/stats3 -name "*kernel_stats.csv" | head -1)" $O/mk3_kernel_stats.txt > /dev/null 2>&1
head -12 $O/mk3_kernel_stats.txt | cut -c1-130
