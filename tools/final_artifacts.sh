#!/bin/bash
# Everything the round's profiles/ directory is made from, in ONE GPU call:  tools/final_artifacts.sh <tag>
#   bench line (default command), rocprofv3 kernel stats of the same command and of the headline-only run, three PMC passes,
#   the mk-3 profile, the BASELINE-config table, the attention bench, the full GPU test log.
# Output: gpurun_out/final_<tag>/ (merged back by gpurun; copied into profiles/ by hand).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r05}
R=$PWD; O=$R/gpurun_out/final_$TAG; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
(rocm-smi --showproductname; rocm-smi --showclocks; uname -a) > $O/${TAG}_box_info.txt 2>&1
timeout 900 python bench.py > $O/${TAG}_bench_final.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-400 $O/${TAG}_bench_final.json
timeout 1500 python -m pytest tests -q -m gpu > $O/${TAG}_pytest_gpu_final.log 2>&1; echo "pytest rc=$?"; tail -3 $O/${TAG}_pytest_gpu_final.log
timeout 600 python -m pytest tests/test_gpu_conv.py -q -s -m gpu -k "end_to_end or config3" 2>&1 | grep "E2E_FP64\|CFG3_FP64\|passed\|failed" > $O/${TAG}_e2e_fp64_errors.txt
timeout 300 python -m pytest tests/test_gpu_baseline_sizes.py -q -s -m gpu -k "contextual" 2>&1 | grep "CTX_FP64\|passed\|failed" | cut -c1-600 > $O/${TAG}_contextual_fp64_errors.txt
timeout 300 python tools/contextual_bench.py 2>/dev/null | grep "^{" > $O/${TAG}_contextual_bench.txt
timeout 600 python tools/configs_bench.py > $O/configs_bench.log 2>&1; cp gpurun_out/configs_bench.json $O/${TAG}_configs_bench.json 2>/dev/null
timeout 300 python tools/attention_bench.py > $O/${TAG}_attention_bench.txt 2>&1; tail -3 $O/${TAG}_attention_bench.txt | cut -c1-300
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1
python $R/tools/rocprof_summary.py "$(find $O/stats -name "*kernel_stats.csv" | head -1)" $O/${TAG}_bench_kernel_stats.txt > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_h -o b -- python $R/bench.py --no-cpu-baseline --no-extras > $O/bench_headline_under_rocprof.log 2>&1
python $R/tools/rocprof_summary.py "$(find $O/stats_h -name "*kernel_stats.csv" | head -1)" $O/${TAG}_bench_headline_kernel_stats.txt > /dev/null 2>&1
rm -rf $O/stats $O/stats_h
head -14 $O/${TAG}_bench_headline_kernel_stats.txt | cut -c1-170
A="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
timeout 200 rocprofv3 --pmc $A --kernel-trace --output-format csv -d $O/pmc_a -o p -- python $R/tools/kernel_bench.py > $O/pmc_a.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_b -o p -- python $R/tools/kernel_bench.py > $O/pmc_b.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_c -o p -- python $R/tools/kernel_bench.py > $O/pmc_c.log 2>&1
python $R/tools/pmc_to_json.py $O/${TAG}_pmc_f16x3.json $(find $O/pmc_a $O/pmc_b $O/pmc_c -name "*counter_collection.csv") > $O/${TAG}_pmc_f16x3.txt 2>&1
rm -rf $O/pmc_a $O/pmc_b $O/pmc_c
cut -c1-260 $O/${TAG}_pmc_f16x3.txt | grep "corr_\|hgemm" | head
cd $R; bash tools/profile_mk3.sh $TAG > $O/profile_mk3.log 2>&1
cp gpurun_out/prof_${TAG}_mk3/${TAG}_mk3_kernel_stats.txt gpurun_out/prof_${TAG}_mk3/${TAG}_mk3_pmc.json gpurun_out/prof_${TAG}_mk3/${TAG}_mk3_pmc.txt $O/ 2>/dev/null
rm -rf gpurun_out/prof_${TAG}_mk3/pmc_* gpurun_out/prof_${TAG}_mk3/stats
head -12 $O/${TAG}_mk3_kernel_stats.txt | cut -c1-170
