#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd); O=$R/gpurun_out/r6_g; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
for mk in 1 3; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats$mk -o b -- python $R/tools/step_bench.py --iters 30 --match-kernel $mk > $O/log$mk.txt 2>&1
python $R/tools/rocprof_summary.py "$(find $O/stats$mk -name "*kernel_stats.csv" | head -1)" $O/mk${mk}_kernel_stats.txt > /dev/null 2>&1
grep -E "proj_|center_l2|absmax|unfold3|elementwise|split_f16" $O/mk${mk}_kernel_stats.txt | cut -c1-140
echo
done
