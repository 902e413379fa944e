#!/bin/bash
# rocprofv3 kernel stats + PMC passes of the K16b kernels at the module's layer shapes (tools/conv_nhwc_bench.py):
#   tools/profile_k16b.sh <tag>   -> gpurun_out/prof_<tag>_k16b/{<tag>_k16b_kernel_stats.txt, <tag>_k16b_pmc.json, .txt}
# Counter passes are separate runs with --kernel-trace only (MI355X guide: never combine --pmc with other trace domains).
set -u
TAG=${1:-r03}
R=$(pwd); O=$R/gpurun_out/prof_${TAG}_k16b; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python $R/tools/conv_nhwc_bench.py > $O/stats.log 2>&1
python $R/tools/rocprof_summary.py "$(find $O/stats -name "*kernel_stats.csv" | head -1)" $O/${TAG}_k16b_kernel_stats.txt > /dev/null 2>&1
A="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
timeout 200 rocprofv3 --pmc $A --kernel-trace --output-format csv -d $O/pmc_a -o p -- python $R/tools/conv_nhwc_bench.py > $O/pmc_a.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_b -o p -- python $R/tools/conv_nhwc_bench.py > $O/pmc_b.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_c -o p -- python $R/tools/conv_nhwc_bench.py > $O/pmc_c.log 2>&1
python $R/tools/pmc_to_json.py $O/${TAG}_k16b_pmc.json $(find $O/pmc_a $O/pmc_b $O/pmc_c -name "*counter_collection.csv") > $O/${TAG}_k16b_pmc.txt 2>&1
rm -rf $O/stats $O/pmc_a $O/pmc_b $O/pmc_c
head -14 $O/${TAG}_k16b_kernel_stats.txt | cut -c1-170
cat $O/${TAG}_k16b_pmc.txt | cut -c1-330
