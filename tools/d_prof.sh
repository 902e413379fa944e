#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/d_prof; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
python tools/d_prof.py 2>&1 | tail -60 | cut -c1-400
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python $R/tools/d_prof.py > $O/log 2>&1
python $R/tools/rocprof_summary.py "$(find $O/stats -name "*kernel_stats.csv" | head -1)" $O/d_kernel_stats.txt > /dev/null 2>&1; rm -rf $O/stats
head -30 $O/d_kernel_stats.txt | cut -c1-190
