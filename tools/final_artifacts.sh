cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/final2; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-300 $O/bench_default.json
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1
python $R/tools/rocprof_summary.py "$(find $O/stats -name "*kernel_stats.csv" | head -1)" $O/r02_bench_kernel_stats.txt > /dev/null 2>&1
head -22 $O/r02_bench_kernel_stats.txt | cut -c1-160
rm -rf $O/stats/*/*kernel_trace.csv
