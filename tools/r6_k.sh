#!/bin/bash
# round 6, call K: K25 (match_kernel 3's projections fused: planes + sums, no fp32 projection) — tests, smoke, A/B, kernel stats
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd); O=$R/gpurun_out/r6_k; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_proj_norm.py -q -m gpu -x -k "k25 or lazy" -s > $O/pytest_k25.log 2>&1; echo "k25 rc=$?"; grep -E "K25_|passed|failed|Error|assert" $O/pytest_k25.log | cut -c1-250 | tail -30
timeout 900 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
for rep in 1 2; do for f in 0 1; do
  COCOS_PROJ_RAW_FUSED=$f timeout 300 python tools/step_bench.py --iters 300 --match-kernel 3 2>&1 | grep match_kernel | sed "s/^/raw_fused=$f /"
done; done
export TMPDIR=/tmp; cd /tmp
for mk in 3; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats$mk -o b -- python $R/tools/step_bench.py --iters 30 --match-kernel $mk > $O/log$mk.txt 2>&1
python $R/tools/rocprof_summary.py "$(find $O/stats$mk -name "*kernel_stats.csv" | head -1)" $O/mk${mk}_kernel_stats.txt > /dev/null 2>&1
head -30 $O/mk${mk}_kernel_stats.txt | cut -c1-130
done
