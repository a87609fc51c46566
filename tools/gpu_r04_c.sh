#!/usr/bin/env bash
# Round 4, visit C: gemm_pp A/B runs (zero C operand; K = N = 640 dispatch) + the kernel tests that hold gemm_pp's bit-identity
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
timeout 600 python tools/pp_zero_c_ab.py > "$out/r04_pp_zero_c_ab.txt" 2>&1
echo "ab rc=$?"; cat "$out/r04_pp_zero_c_ab.txt"
timeout 900 python -m pytest -q -m gpu -p no:cacheprovider tests/test_gpu_ops.py tests/test_gpu_ln_fusion.py -k "gemm or ln_fusion or pingpong" > "$out/r04_pytest_pp.log" 2>&1
echo "pytest rc=$?"; tail -4 "$out/r04_pytest_pp.log"
