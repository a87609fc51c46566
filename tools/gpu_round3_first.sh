#!/usr/bin/env bash
# First GPU call of the next round: the experimental 4-wave GEMM (csrc/experimental/gemm_w4.hip) meets a GPU.
# Build the experimental library BEFORE calling gpurun (the .so travels with the snapshot):
#     python -m diffsensei_amd.build --experimental --force
#     gpurun --timeout 600 -- 'bash tools/gpu_round3_first.sh'
#     python -m diffsensei_amd.build --force          # back to the production library
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
# 1. parity: bit-equality with the register-staged kernel (skips if the library is the production one)
timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "w4" 2>&1 | grep -E "passed|failed|skipped|error" | tail -2
# 2. timing against the ping-pong kernel and F.linear, UNet level-2 shapes
timeout 300 python tools/w4_check.py 2> "$out/r03_w4_check.err" | tee "$out/r03_w4_check.txt"
tail -2 "$out/r03_w4_check.err"
