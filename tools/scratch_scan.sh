#!/bin/bash
# Every kernel of the library whose code object uses scratch (register spills) — compile-only, no GPU:  tools/scratch_scan.sh
cd "$(dirname "$0")/.."
O=/tmp/cocos_scratch_scan; rm -rf $O; mkdir -p $O
ls cocosnet_amd/csrc/*.hip | xargs -P 8 -I{} sh -c 'b=$(basename {} .hip); X=""; [ $b = proj_dw_f16x3 ] && X="-mllvm -amdgpu-mfma-vgpr-form"; hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-function $X --offload-device-only --no-gpu-bundle-output -c {} -o '$O'/$b.co 2>/dev/null'
python3 - $O <<'PY'
import subprocess, re, glob, sys
tot = nk = 0
for f in sorted(glob.glob(sys.argv[1] + '/*.co')):
    out = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-readelf', '--notes', f], capture_output=True, text=True).stdout
    names = re.findall(r"\.name:\s+(\S+)", out); ps = re.findall(r"\.private_segment_fixed_size:\s+(\d+)", out); vg = re.findall(r"\.vgpr_count:\s+(\d+)", out)
    for n, p, v in zip(names, ps, vg):
        nk += 1
        if int(p) > 0:
            print(f.split('/')[-1][:-3], n[:110], 'scratch', p, 'B/lane, vgpr', v); tot += 1
print('kernels:', nk, 'with scratch:', tot)
PY
