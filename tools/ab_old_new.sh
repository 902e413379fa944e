#!/bin/bash
# same-box A/B of the default bench: the tree under _ab_old/ (an earlier commit, own library) vs this tree
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
for rep in 1 2; do
for d in _ab_old .; do
  cd $R/$d
  echo -n "$d: "; timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], {k:v['avg_ms'] for k,v in d['kernels'].items()})"
done; done
