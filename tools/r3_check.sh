#!/bin/bash
# Round-3 checkpoint in one GPU call: full GPU suite, smoke, netcorr scope in the three convolution flavours.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/r3_check; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
for f in f16x3 bf16 torch; do
  COCOS_CONV=$f timeout 400 python bench.py --scope netcorr --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/netcorr_$f.json
  echo "netcorr $f: $(cut -c1-160 $O/netcorr_$f.json)"
done
