#!/bin/bash
# round 6, call E: kernel stats + one step's ordered kernel trace of the match_kernel-3 bench step
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd); O=$R/gpurun_out/r6_e; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python $R/tools/step_bench.py --iters 20 --match-kernel 3 > $O/log.txt 2>&1
tail -2 $O/log.txt
python $R/tools/rocprof_summary.py "$(find $O/stats -name "*kernel_stats.csv" | head -1)" $O/mk3_kernel_stats.txt > /dev/null 2>&1
head -45 $O/mk3_kernel_stats.txt | cut -c1-150
python - <<PY
import csv,glob
f=glob.glob("$O/stats/*kernel_trace.csv")[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last step: find the last occurrence of the first kernel of a step (absmax after upsample?) -> print the last 70 kernels
tail=rows[-75:]
t0=int(tail[0]["Start_Timestamp"])
for r in tail:
    print(f'{(int(r["Start_Timestamp"])-t0)/1e3:9.1f} {(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:8.1f}  {r["Kernel_Name"][:90]}')
PY
