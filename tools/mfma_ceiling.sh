#!/bin/bash
# the sustained f16 MFMA ceiling with sclk / power sampled beside it -> gpurun_out/mfma_ceiling/
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/mfma_ceiling; mkdir -p $O
[ -x tools/probes/mfma_ceiling ] || hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_ceiling.hip -o tools/probes/mfma_ceiling
( while true; do date +%s.%N; rocm-smi --showclocks --showpower 2>&1 | grep -iE "sclk|power|mclk|fclk"; sleep 0.2; done ) > $O/smi_samples.txt 2>&1 &
SMI=$!
sleep 1
tools/probes/mfma_ceiling 3 1 > $O/probe_1wg.txt 2>&1
sleep 1
tools/probes/mfma_ceiling 3 2 > $O/probe_2wg.txt 2>&1
kill $SMI
rocm-smi --showclocks --showpower --showperflevel --showmaxpower > $O/smi_idle_after.txt 2>&1
cat $O/probe_1wg.txt; cat $O/probe_2wg.txt; grep -c . $O/smi_samples.txt; head -30 $O/smi_samples.txt
