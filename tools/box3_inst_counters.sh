#!/bin/bash
# round 6, call P: instruction-mix counters of the match_kernel-3 kernels
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd); O=$R/gpurun_out/r6_p; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $O/pmc -o p -- python $R/tools/step_bench.py --iters 3 --match-kernel ${MK:-3} > $O/log.txt 2>&1
python - <<PY
import csv,glob,collections
f=glob.glob("$O/pmc/*counter_collection.csv")[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"][:60]
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
    if r["Counter_Name"]=="SQ_WAVES": cnt[k]+=1
for k,v in agg.items():
    if True:
        n=cnt[k] or 1; w=v["SQ_WAVES"]/n
        print(k, "launches",n, {c: round(x/n/max(w,1)) for c,x in v.items() if c!="SQ_WAVES"}, "waves", round(w))
PY
