#!/usr/bin/env bash
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
timeout 600 python -m pytest -q -m gpu -p no:cacheprovider -x tests/test_gpu_ops.py -k "row640 or gemm_bias" 2>&1 | tail -8
timeout 300 python tools/gemm_row_ab.py > "$out/r04_gemm_row640_ab.txt" 2>&1
echo "rc=$?"; cat "$out/r04_gemm_row640_ab.txt"
