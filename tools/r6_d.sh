#!/bin/bash
# round 6, call D: K23 tests again (transposed planes now come from the prep kernel), A/B, and PMC of EVERY kernel of the bench step
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd); O=$R/gpurun_out/r6_d; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_proj_norm.py tests/test_gpu_baseline_sizes.py -q -m gpu -x -k "k23 or lazy or bench_configuration" > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -4 $O/pytest.log
for f in 0 1 0 1; do
  COCOS_PROJ_NORM_FUSED=$f timeout 300 python tools/step_bench.py --iters 300 2>&1 | sed "s/^/fused=$f /"
done
export TMPDIR=/tmp; cd /tmp
A="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
B2="SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE"
timeout 300 rocprofv3 --pmc $A --kernel-trace --output-format csv -d $O/pmc_a -o p -- python $R/tools/step_bench.py --iters 4 > $O/pmc_a.log 2>&1
timeout 300 rocprofv3 --pmc $B2 --kernel-trace --output-format csv -d $O/pmc_a2 -o p -- python $R/tools/step_bench.py --iters 4 > $O/pmc_a2.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_b -o p -- python $R/tools/step_bench.py --iters 4 > $O/pmc_b.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_c -o p -- python $R/tools/step_bench.py --iters 4 > $O/pmc_c.log 2>&1
python $R/tools/pmc_to_json.py $O/r06_step_pmc.json $(find $O/pmc_a $O/pmc_a2 $O/pmc_b $O/pmc_c -name "*counter_collection.csv") > $O/r06_step_pmc.txt 2>&1
cat $O/r06_step_pmc.txt | cut -c1-700
