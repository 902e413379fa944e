#!/bin/bash
# Round-2 GPU call A: quick parity of the restructured K2 kernels, same-box A/B against the round-1 tree, full GPU
# test suite, rocprof kernel stats.  Everything lands under gpurun_out/r2a/.
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r2a; mkdir -p $O
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -m1 "Marketing Name" > $O/box.txt; nproc >> $O/box.txt
echo "== quick parity" ; timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "fused_forward_backward_vs_oracle or theta_phi_only or reference_fixtures or hgemm" > $O/quick.log 2>&1; echo "quick rc=$?" ; tail -15 $O/quick.log
echo "== bench new"; timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > $O/bench_new.json 2> $O/bench_new.err; echo "rc=$?"; tail -c 600 $O/bench_new.err
echo "== bench r1"; (cd _r1_baseline && timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > ../$O/bench_r1.json 2> ../$O/bench_r1.err); echo "rc=$?"
echo "== bench new again (ordering effects)"; timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extras > $O/bench_new2.json 2>> $O/bench_new.err
python - <<'PY'
import json
for f in ("bench_new","bench_r1","bench_new2"):
    try:
        d=json.loads(open(f"gpurun_out/r2a/{f}.json").read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], {k:v["avg_ms"] for k,v in d["kernels"].items()}, d.get("stability"), {k:v["ms_per_step"] for k,v in (d.get("flavours") or {}).items()})
        print("   ", d["abi_calls_ms_per_step"])
    except Exception as e: print(f, "ERR", e)
PY
echo "== full gpu tests"; timeout 1800 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "rc=$?"; tail -30 $O/pytest_gpu.log
echo "== rocprof"; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/rocprof.log 2>&1; echo "rc=$?"
cd $GRAFT_REPO_ROOT; F=$(find $O/prof -name "*.db" -o -name "*kernel_stats.csv" | head -1); echo "prof file: $F"; python tools/rocprof_summary.py "$F" $O/kernel_stats.txt > /dev/null 2>&1; head -30 $O/kernel_stats.txt
