#!/bin/bash
# round 6, call B: K23 (projection fused with K1) — its tests, the bench-configuration fp64 tests, a short bench with per-call timings
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r6_b; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_proj_norm.py -q -m gpu -x -s > $O/pytest_k23.log 2>&1; echo "k23 rc=$?"; tail -25 $O/pytest_k23.log
timeout 900 python -m pytest tests/test_gpu_baseline_sizes.py -q -m gpu -x -k "bench_configuration" > $O/pytest_bench_cfg.log 2>&1; echo "benchcfg rc=$?"; tail -8 $O/pytest_bench_cfg.log
timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extras > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
try:
    d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
    print("ms/step", d["ms_per_step"], "stability", d.get("stability",{}).get("ms_per_step"), "frac", d["roofline"]["frac"])
    print(d["abi_calls_ms_per_step"])
except Exception as e: print("bench parse", e); print(open("$O/bench.err").read()[-3000:])
PY
