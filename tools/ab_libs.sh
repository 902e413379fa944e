#!/bin/bash
# same-box A/B of the default bench between the product library and alternate builds:  tools/ab_libs.sh lib1.so lib2.so ...
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for rep in 1 2; do
for L in "$@"; do
  echo -n "$L: "; COCOS_LIB_PATH=$PWD/cocosnet_amd/lib/$L timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], {k:v['avg_ms'] for k,v in d['kernels'].items()})"
done; done
