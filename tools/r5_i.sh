#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5_i; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5_i/bench.json"))
r=d["roofline"]
print(d["value"], d["ms_per_step"], r["kernel"], r["frac"], r["avg_launch_ms"], d["config"]["hip_events_in_timed_window"], d["stability"]["ms_per_step"])
c=d["config"]["context"]
print("mk3", c["match_kernel_3"]["ms_per_step"], "module", c["module_scope"]["f16x3"]["ms_per_step"], "cfg3", c["config3"]["f16x3"]["generator"], c["config3"]["f16x3"]["discriminator"])
PY
