#!/bin/bash
# round 6, call J: dw pair launch, warp head v2, mk3 quick wins — tests, then same-box A/B of the pair launch
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd); O=$R/gpurun_out/r6_j; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_plane_prep.py tests/test_gpu_proj_norm.py tests/test_gpu_parity.py tests/test_gpu_baseline_sizes.py -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
for rep in 1 2; do for f in 0 1; do
  COCOS_PROJ_DW_PAIR=$f timeout 300 python tools/step_bench.py --iters 300 --match-kernel 1 2>&1 | grep match_kernel | sed "s/^/dw_pair=$f /"
done; done
timeout 300 python tools/step_bench.py --iters 300 --match-kernel 3 2>&1 | grep match_kernel
export TMPDIR=/tmp; cd /tmp
for mk in 1 3; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats$mk -o b -- python $R/tools/step_bench.py --iters 30 --match-kernel $mk > $O/log$mk.txt 2>&1
python $R/tools/rocprof_summary.py "$(find $O/stats$mk -name "*kernel_stats.csv" | head -1)" $O/mk${mk}_kernel_stats.txt > /dev/null 2>&1
head -28 $O/mk${mk}_kernel_stats.txt | cut -c1-130
done
