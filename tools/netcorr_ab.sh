#!/bin/bash
# netcorr scope with K16 vs the framework's convolutions, same box; then a kernel trace of the K16 run's steady state
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/netcorr; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
COCOS_CONV=f16x3 timeout 300 python bench.py --scope netcorr --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_k16.json; cut -c1-300 $O/bench_k16.json
COCOS_CONV=torch timeout 400 python bench.py --scope netcorr --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_torch.json; cut -c1-300 $O/bench_torch.json

cd /tmp
COCOS_CONV=f16x3 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python $GRAFT_REPO_ROOT/bench.py --scope netcorr --steps 10 --warmup 3 --no-cpu-baseline > $O/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
ls tools/trace_window_stats.py >/dev/null 2>&1 && python tools/trace_window_stats.py "$(find $O/stats -name '*kernel_trace.csv' | head -1)" 0.25 > $O/steady_state.txt 2>&1
head -40 $O/steady_state.txt
