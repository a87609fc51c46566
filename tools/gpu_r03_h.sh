#!/usr/bin/env bash
# Round 3, GPU call H: sp attention row sums packed (variant 3) vs plain adds (variant 4); sp tests.
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "self_attention_s or sp_" 2>&1 | tail -3
VARS=3,4 ROUNDS=7 timeout 300 python tools/attn_bench.py 2>&1 | grep "^B=" | head -4 | tee "$out/r03_h_attn_pk_ab.txt"
