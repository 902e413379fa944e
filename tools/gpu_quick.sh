#!/bin/bash
# a subset of the GPU tests (-k expression in $2) + optional extra command ($3); logs under gpurun_out/$1
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/${1:-q}; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x --maxfail=50 -k "${2:-test}" > $O/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)" $O/pytest.log | sed 's/ - .*//' | head -40
grep -E "^E  " $O/pytest.log | sort | uniq -c | sort -rn | cut -c1-400 | head -25
tail -2 $O/pytest.log
if [ -n "${3:-}" ]; then timeout 900 bash -c "$3" > $O/extra.log 2>&1; echo "extra rc=$?"; tail -40 $O/extra.log | cut -c1-600; fi
