#!/bin/bash
# instructions between consecutive MFMAs of one basic block: isa_mfma_gaps.sh file.s <kernel substring> <.LBB label>
awk -v k="$2" -v l="$3" 'index($0,k)&&/^_Z/{f=1} f&&index($0,l":")==1{g=1} g{print} g&&/s_cbranch/{exit}' "$1" | grep -v "^\s*;" | awk '{op=$1; if (op ~ /v_mfma/) {printf "%d(v%d s%d m%d l%d w%d) ", n, v, s, m, l, w; n=0;v=0;s=0;m=0;l=0;w=0; c++; if (c%6==0) printf "\n"} else if (op !~ /^\./) {n++; if (op ~ /^v_/) v++; else if (op ~ /s_waitcnt/) w++; else if (op ~ /^s_/) s++; else if (op ~ /^buffer/) m++; else if (op ~ /^ds_/) l++;}} END{printf "tail %d\n", n}'
