#!/usr/bin/env bash
# Round 2, GPU call F: arbitrary latent sizes (resize-upsample conv, ragged attention, padded VAE attention), the whole
# GPU suite, and the default bench with the even-rounds gemm_pp grid + ip_attn block skipping.
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu --durations=5 2>&1 | tail -25 | tee "$out/r02_pytest_gpu_f.log"
timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2> "$out/r02_bench_ns16_f.err" \
    | tail -1 | tee "$out/r02_bench_ns16_f.json" | cut -c1-300
DS_OPTIONS=gemm_pp_even=0 timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2> "$out/r02_bench_ns16_f_noeven.err" \
    | tail -1 | tee "$out/r02_bench_ns16_f_noeven.json" | cut -c1-200
