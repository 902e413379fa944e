#!/bin/bash
# K22 (contextual_fused_f16x3.hip) under rocprofv3: kernel stats + the three PMC passes of the guide (SQ_* + GRBM; FETCH_SIZE; WRITE_SIZE)
# at B = 8, C = 512, N = 4096 forward + backward w.r.t. both sides.  Output: gpurun_out/ctxpmc/r05_contextual_pmc.{json,txt}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/ctxpmc; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
cat > $O/drv.py <<'PY'
import sys, torch
sys.path.insert(0, sys.argv[1])
from cocosnet_amd import ops
B, C, N = 8, 512, 4096
g = torch.Generator(device="cuda").manual_seed(1)
Y = torch.randn(B, C, N, device="cuda", generator=g)
X = 0.6 * Y[:, :, torch.randperm(N, device="cuda", generator=g)] + torch.randn(B, C, N, device="cuda", generator=g)
nrm = lambda t: (t / (t.norm(dim=1, keepdim=True) + 2.2e-16)).contiguous()
Xn, Yn = nrm(X), nrm(Y)
for _ in range(4):
    x, y = Xn.clone().requires_grad_(True), Yn.clone().requires_grad_(True)
    ops.contextual_cx(x, y, 0.1, 1e-3).sum().backward()
torch.cuda.synchronize()
PY
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python $O/drv.py $R > $O/stats.log 2>&1
python $R/tools/rocprof_summary.py "$(find $O/stats -name "*kernel_stats.csv" | head -1)" $O/r05_contextual_kernel_stats.txt > /dev/null 2>&1
A="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
timeout 200 rocprofv3 --pmc $A --kernel-trace --output-format csv -d $O/pmc_a -o p -- python $O/drv.py $R > $O/pmc_a.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_b -o p -- python $O/drv.py $R > $O/pmc_b.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_c -o p -- python $O/drv.py $R > $O/pmc_c.log 2>&1
python $R/tools/pmc_to_json.py $O/r05_contextual_pmc.json $(find $O/pmc_a $O/pmc_b $O/pmc_c -name "*counter_collection.csv") > $O/r05_contextual_pmc.txt 2>&1
rm -rf $O/stats $O/pmc_a $O/pmc_b $O/pmc_c
head -8 $O/r05_contextual_kernel_stats.txt | cut -c1-200
grep "cf_kernel" $O/r05_contextual_pmc.txt | cut -c1-420
