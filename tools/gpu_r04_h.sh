#!/usr/bin/env bash
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
timeout 300 python tools/l1_variants_ab.py > "$out/r04_l1_variants_ab.txt" 2>&1
echo "rc=$?"; cat "$out/r04_l1_variants_ab.txt"
