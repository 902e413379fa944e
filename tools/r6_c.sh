#!/bin/bash
# round 6, call C: same-box A/B of the headline step with K23 on / off (twice each, interleaved), and rocprofv3 kernel stats of the K23 run
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd); O=$R/gpurun_out/r6_c; rm -rf $O; mkdir -p $O
for i in 1 2; do
  for f in 0 1; do
    COCOS_PROJ_NORM_FUSED=$f timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --stability-steps 300 > $O/bench_f${f}_$i.json 2> $O/bench_f${f}_$i.err
    python - <<PY
import json
d=json.loads(open("$O/bench_f${f}_$i.json").read().strip().splitlines()[-1])
print("fused=$f run $i: ms/step", d["ms_per_step"], "stability", (d.get("stability") or {}).get("ms_per_step"), "frac", d["roofline"]["frac"])
PY
  done
done
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python $R/bench.py --no-cpu-baseline --no-extras > $O/bench_under_rocprof.log 2>&1
python $R/tools/rocprof_summary.py "$(find $O/stats -name "*kernel_stats.csv" | head -1)" $O/r06_bench_k23_kernel_stats.txt > /dev/null 2>&1
head -40 $O/r06_bench_k23_kernel_stats.txt
