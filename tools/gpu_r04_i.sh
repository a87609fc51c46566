#!/usr/bin/env bash
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
timeout 600 python tools/forward_option_ab.py 64 ip_attn_variant=1,2,0 gn_variant=1,0 gemm_debug=2048,0 > "$out/r04_forward_option_ab.txt" 2>&1
echo "rc=$?"; cat "$out/r04_forward_option_ab.txt"
