#!/usr/bin/env bash
# Round 3, GPU call C: the software-pipelined self-attention kernel (attn_variant 3): parity tests, then timing vs the default.
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "self_attention_s or sp_" 2>&1 | tail -15 | tee "$out/r03_c_pytest.log"
timeout 300 python tools/attn_bench.py 2>&1 | tail -12 | tee "$out/r03_c_attn_bench.txt"
