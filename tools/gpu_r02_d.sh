#!/usr/bin/env bash
# Round 2, GPU call D: fp8 attention - parity tests, microbench vs fp16, BASELINE config 5 shape (2048^2, 4 refs, num_samples 1)
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_attention_fp8.py tests/test_gpu_ops.py -q -m gpu -x -s 2>&1 | grep -v "^$" | tail -25 | tee "$out/r02_pytest_fp8_d.log"
timeout 300 python tools/attn_fp8_bench.py 2>&1 | grep -v amdgpu.ids | tee "$out/r02_attn_fp8_bench.txt"
for a in fp16 fp8; do
timeout 500 python bench.py --size 2048 --refs 4 --num-samples 1 --attn $a --steps 2 --warmup 1 --no-cpu-baseline 2> "$out/r02_bench_c5_$a.err" \
    | tail -1 | tee "$out/r02_bench_c5_2048_ns1_$a.json" | cut -c1-300
done
