#!/bin/bash
# same-box A/B of the bench step: round 5's final tree (_ab_old/, commit 19aaa83, its own library) vs this tree — match_kernel 1 and 3,
# interleaved, twice; `stability` = 300 untimed-contract steps without HIP events
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
for mk in 1 3; do for rep in 1 2; do for d in _ab_old .; do
  cd $R/$d
  echo -n "match_kernel $mk  $([ $d = . ] && echo round6 || echo round5): "
  timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --match-kernel $mk --stability-steps 300 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'stability', (d.get('stability') or {}).get('ms_per_step'))"
done; done; done
