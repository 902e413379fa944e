#!/bin/bash
# PMC passes over the K0 streaming kernels (tools/proj_bench.py at the benchmark shape 8 x 407 -> 256 x 64 x 64)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/k0pmc; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp COCOS_BENCH_NO_TORCH=1; cd /tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_WAVES" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o p -- python $R/tools/proj_bench.py > $O/p$i.log 2>&1
  f=$(find $O/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(float); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "proj_" not in k: continue
    k = k.split("(")[0].replace("cocos::", "")[:40]
    agg[(k, r["Counter_Name"])] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for (k, c), v in sorted(agg.items()): print(f"{k:42s} {c:30s} {v / max(cnt[(k, c)], 1):16.1f}  ({cnt[(k, c)]} launches)")
PY
  tail -2 $O/p$i.log | grep -i "error\|invalid" | head -2
done
