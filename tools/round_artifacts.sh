#!/bin/bash
# Round artefacts in one GPU call:  tools/round_artifacts.sh <tag>   (outputs under gpurun_out/art_<tag>/)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; T=${1:-r02}; O=$R/gpurun_out/art_$T; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-400 $O/bench_default.json
timeout 300 python tools/configs_bench.py > $O/configs_bench.json 2> $O/configs_bench.err; echo "configs rc=$?"; tail -12 $O/configs_bench.json | cut -c1-300
timeout 300 python tools/precision_check.py > $O/precision_check.txt 2>&1; echo "precision rc=$?"; tail -8 $O/precision_check.txt | cut -c1-300
rocm-smi --showproductname --showclocks 2>/dev/null | head -30 > $O/box_info.txt
tools/profile_round.sh $T > $O/profile_round.log 2>&1; echo "profile rc=$?"; tail -30 $O/profile_round.log | cut -c1-300
cp -r $R/gpurun_out/prof_$T/${T}_* $O/ 2>/dev/null
