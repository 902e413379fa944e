#!/bin/bash
# full GPU test suite + short bench; logs under gpurun_out/$1
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/${1:-t}; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x --maxfail=${2:-400} > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)" $O/pytest_gpu.log | sed 's/ - .*//' | head -60
grep -E "^E  " $O/pytest_gpu.log | sort | uniq -c | sort -rn | head -25
tail -3 $O/pytest_gpu.log
timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extras > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
try:
    d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
    print("ms/step", d["ms_per_step"], {k:v["avg_ms"] for k,v in d["kernels"].items()}, "frac", d["roofline"]["frac"])
    print(d["abi_calls_ms_per_step"])
except Exception as e: print("bench parse", e)
PY
