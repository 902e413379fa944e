#!/bin/bash
# The three PMC passes over the module's big convolution shapes in the default flavour (K16c: f16 hi/lo planes NHWC, three MFMA terms):
#   tools/conv_f16x3_pmc.sh <tag>     (tools/conv_nhwc_bench.py with PREC=f16x3)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r06}; R=$PWD; O=$R/gpurun_out/conv_pmc_$TAG; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp PREC=f16x3
cd /tmp
A="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
timeout 300 rocprofv3 --pmc $A --kernel-trace --output-format csv -d $O/pmc_a -o p -- python $R/tools/conv_nhwc_bench.py > $O/pmc_a.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_b -o p -- python $R/tools/conv_nhwc_bench.py > $O/pmc_b.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_c -o p -- python $R/tools/conv_nhwc_bench.py > $O/pmc_c.log 2>&1
python $R/tools/pmc_to_json.py $O/${TAG}_conv_f16x3_pmc.json $(find $O/pmc_a $O/pmc_b $O/pmc_c -name "*counter_collection.csv") > $O/${TAG}_conv_f16x3_pmc.txt 2>&1
rm -rf $O/pmc_a $O/pmc_b $O/pmc_c
grep "conv_nhwc" $O/${TAG}_conv_f16x3_pmc.txt | cut -c1-330
