#!/bin/bash
# Kernel trace of the netcorr scope's steady state in one convolution flavour:  tools/netcorr_prof.sh [bf16|f16x3|torch]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; F=${1:-bf16}; O=$R/gpurun_out/netcorr_prof_$F; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
COCOS_CONV=$F timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python $R/bench.py --scope netcorr --steps 10 --warmup 3 --no-cpu-baseline > $O/rocprof.log 2>&1
cd $R
python tools/trace_window_stats.py "$(find $O/stats -name '*kernel_trace.csv' | head -1)" 0.25 > $O/steady_state.txt 2>&1
rm -rf $O/stats
head -70 $O/steady_state.txt
