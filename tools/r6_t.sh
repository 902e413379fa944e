#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r6_t; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_mk3_sizes.py -q -m gpu -k "t_storage" 2>&1 | tail -3
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real
python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "steps", d["steps"], "stability", d["stability"]["ms_per_step"])
r=d["roofline"]; print({k:r[k] for k in ("kernel","achieved","frac","avg_launch_ms","frac_issued","frac_of_sustained")})
print(d["config"]["hip_events_in_timed_window"], d["config"]["context"]["match_kernel_3"]["ms_per_step"])
PY
