#!/bin/bash
# round 6, call N: the driver's bench line (default flags) + a longer one
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r6_n; rm -rf $O; mkdir -p $O
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "stability", (d.get("stability") or {}).get("ms_per_step"))
r=d["roofline"]; print("roofline", r["kernel"], r["frac"], r["avg_launch_ms"], r.get("frac_of_sustained"))
print("abi", d["abi_calls_ms_per_step"])
c=d["config"]["context"]
print("mk3", c["match_kernel_3"]["ms_per_step"])
print("mk3 abi", c["match_kernel_3"]["abi_calls_ms_per_step"])
for k in ("config5","config3","module_scope"):
    v=c.get(k); print(k, json.dumps(v)[:600])
print("cpu", json.dumps(d["cpu_baseline"])[:300])
PY
tail -3 $O/bench_default.err
