#!/bin/bash
# Debug: SQ counters of the K2 split forward in its two flavours (magnitude-free / unit-norm) at the same geometry
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/attnpmc; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_WAVES" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  for m in rawm unit; do
    timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i$m -o p -- python $R/tools/attention_fwd_only.py $m > $O/p$i$m.log 2>&1
    f=$(find $O/p$i$m -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python - "$f" $m <<'PY'
import csv, sys, collections
agg = collections.defaultdict(float); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "corr_fwd_f16x3" not in k: continue
    agg[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
for c, v in sorted(agg.items()): print(f"{sys.argv[2]:5s} {c:30s} {v / max(cnt[c], 1):16.1f}  ({cnt[c]} launches)")
PY
    tail -2 $O/p$i$m.log | grep -i "error\|invalid" | head -2
  done
done
