#!/usr/bin/env bash
# Round 2, GPU call C: ring-buffered small-grid GEMM + split-K tail of gemm_pp: parity tests, A/B microbench, whole-call
# numbers at num_samples 1 / 4 (BASELINE configs 2 / 3 shapes).
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "gemm" 2>&1 | tail -15 | tee "$out/r02_pytest_gemm_c.log"
timeout 600 python tools/small_batch_gemm_ab.py 2>&1 | grep -v amdgpu.ids | tee "$out/r02_small_batch_gemm_ab.txt"
timeout 300 python bench.py --num-samples 1 --refs 1 --no-dialog --steps 3 --warmup 1 --no-cpu-baseline 2> "$out/r02_bench_c2_v2.err" \
    | tail -1 | tee "$out/r02_bench_c2_ns1_v2.json" | cut -c1-300
timeout 300 python bench.py --num-samples 4 --steps 2 --warmup 1 --no-cpu-baseline 2> "$out/r02_bench_ns4_v2.err" \
    | tail -1 | tee "$out/r02_bench_ns4_v2.json" | cut -c1-300
