#!/bin/bash
# Profiles for the round (run on the GPU box through gpurun):  tools/profile_round.sh <tag> [precision]
#   1. rocprofv3 --kernel-trace --stats over the default bench.py run   -> profiles/<tag>_bench_kernel_stats.txt
#   2. three --pmc passes (own runs, --kernel-trace only, as the MI355X guide prescribes) over
#      tools/kernel_bench.py                                             -> profiles/<tag>_pmc.json
# Everything is written under gpurun_out/ (merged back by gpurun) and the summaries are copied to profiles/ by hand.
set -u
TAG=${1:-r01}; PREC=${2:-f16x3}
R=$(pwd); O=$R/gpurun_out/prof_$TAG; mkdir -p $O
export TMPDIR=/tmp COCOS_PRECISION=$PREC COCOS_PROJ_PRECISION=$PREC
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1
python $R/tools/rocprof_summary.py "$(find $O/stats -name "*kernel_stats.csv" | head -1)" $O/${TAG}_bench_kernel_stats.txt > /dev/null 2>&1
A="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
timeout 200 rocprofv3 --pmc $A --kernel-trace --output-format csv -d $O/pmc_a -o p -- python $R/tools/kernel_bench.py > $O/pmc_a.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_b -o p -- python $R/tools/kernel_bench.py > $O/pmc_b.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_c -o p -- python $R/tools/kernel_bench.py > $O/pmc_c.log 2>&1
python $R/tools/pmc_to_json.py $O/${TAG}_pmc.json $(find $O/pmc_a $O/pmc_b $O/pmc_c -name "*counter_collection.csv") > $O/${TAG}_pmc.txt 2>&1
tail -3 $O/bench_under_rocprof.log | cut -c1-300
head -20 $O/${TAG}_bench_kernel_stats.txt
cat $O/${TAG}_pmc.txt | cut -c1-400
# 3. per-phase shader-clock ticks of the K2 split kernels (needs the -DCOCOS_DEBUG_TIMING library next to the product one:
#    COCOS_LIB_NAME=libcocos_hip_dbg.so COCOS_EXTRA_HIPFLAGS=-DCOCOS_DEBUG_TIMING python -m cocosnet_amd.build)
if [ -f $R/cocosnet_amd/lib/libcocos_hip_dbg.so ]; then
  cd $R; COCOS_LIB_PATH=$R/cocosnet_amd/lib/libcocos_hip_dbg.so timeout 200 python tools/phase_timing_f16x3.py 154 train > $O/${TAG}_phase_timing.txt 2>&1; cat $O/${TAG}_phase_timing.txt
fi
