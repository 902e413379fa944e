#!/bin/bash
# PMC passes over the K16 forward at the ResidualBlock shape (tools/conv_fwd_ms.py)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/convpmc; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -oE "^\s*(Name|Counter_Name)\s*:\s*\S+|\b(SQ|TA|TCP|TCC|TD|GRBM)_[A-Za-z0-9_]+" | sed 's/.*:\s*//' | sort -u > $O/avail.txt
wc -l $O/avail.txt
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT" \
           "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum FETCH_SIZE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_WAVES"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o p -- python $R/tools/conv_fwd_ms.py > $O/p$i.log 2>&1
  f=$(find $O/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if "conv_fwd" not in r["Kernel_Name"]: continue
    agg[r["Counter_Name"]]["v"] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
for k, v in agg.items(): print(f"{k:42s} {v['v'] / max(cnt[k], 1):16.1f}  (per launch, {cnt[k]} launches)")
PY
  tail -2 $O/p$i.log | grep -i "error\|invalid" | head -2
done
