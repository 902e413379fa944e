#!/bin/bash
# per-kernel times of the K0 streaming kernels (rocprofv3 --kernel-trace --stats over tools/proj_bench.py)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/k0prof; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp COCOS_BENCH_NO_TORCH=1; cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o p -- python $R/tools/proj_bench.py "$@" > $O/log.txt 2>&1
f=$(find $O -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "proj" in r["Name"] or "absmax" in r["Name"] or "split" in r["Name"]:
        print(f'{float(r["AverageNs"])/1e3:8.1f} us  x{r["Calls"]:>5s}  {r["Name"][:90]}')
PY
