#!/bin/bash
# Round-3 final artefacts in one GPU call (outputs under gpurun_out/r3_final/; copy what is judged into profiles/).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r3_final; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench_final.json 2> $O/bench_final.err; echo "bench rc=$?"; cut -c1-300 $O/bench_final.json
timeout 400 python tools/configs_bench.py > $O/configs_bench.log 2>&1; cp gpurun_out/configs_bench.json $O/ 2>/dev/null; tail -6 $O/configs_bench.log | cut -c1-200
for f in f16x3 bf16 torch; do
  COCOS_CONV=$f timeout 400 python bench.py --scope netcorr --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/netcorr_$f.json
  echo "netcorr $f: $(python -c "import json;d=json.load(open('$O/netcorr_$f.json'));print(d['ms_per_step'], d['value'])")"
done
timeout 300 python tools/conv_nhwc_bench.py > $O/conv_nhwc_bench.jsonl 2>/dev/null
timeout 300 python tools/conv_nhwc_gemm_bench.py > $O/conv_nhwc_gemm_bench.jsonl 2>/dev/null
timeout 300 python tools/instnorm_bench.py > $O/instnorm_bench.jsonl 2>/dev/null
COCOS_CONV=bf16 timeout 300 python tools/conv_shapes_netcorr.py > $O/conv_shapes_bf16.txt 2>&1
tools/netcorr_prof.sh bf16 > /dev/null 2>&1; cp gpurun_out/netcorr_prof_bf16/steady_state.txt $O/netcorr_bf16_steady_state.txt 2>/dev/null
rocm-smi --showproductname --showclocks 2>/dev/null | head -30 > $O/box_info.txt
ls $O
