#!/usr/bin/env bash
# Round 2, GPU call L: RMSNorm gains applied in the GEMV prologue (reference roundings): MLLM tests, token-loop rate, config 3
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_mllm.py tests/test_gpu_pipeline_variants.py -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3 | tee "$out/r02_pytest_l.log"
timeout 300 python tools/mllm_bench.py --graph on --new 192 2>/dev/null | tail -1 | tee "$out/r02_mllm_decode_bench.json" | cut -c1-400
