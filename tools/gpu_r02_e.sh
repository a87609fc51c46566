#!/usr/bin/env bash
# Round 2, GPU call E: ip_attn block skipping (parity + time), fp8 attention tests after the tolerance rewrite
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_attention_fp8.py tests/test_gpu_ops.py -q -m gpu -s 2>&1 | grep -E "passed|failed|rel-L2|Error|assert" | tail -25 | tee "$out/r02_pytest_e.log"
timeout 200 python tools/ipattn_bench.py 2>&1 | grep -v amdgpu.ids | tee "$out/r02_ipattn_bench.txt"
timeout 200 python tools/pp_even_ab.py 2>&1 | grep -v amdgpu.ids | tee "$out/r02_pp_even_ab.txt"
