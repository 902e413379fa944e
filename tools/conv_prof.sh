#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/convprof; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp; R=$PWD
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o c -- python $R/tools/conv_bench.py 8 > $O/log.txt 2>&1
python $R/tools/rocprof_summary.py "$(find $O -name '*kernel_stats.csv' | head -1)" $O/summary.txt > /dev/null 2>&1
head -30 $O/summary.txt | cut -c1-200
