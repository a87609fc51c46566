#!/usr/bin/env bash
# Round 2, GPU call B: the new parity tests (halo256 / self_attn<2> / SDXL-shape forward vs oracle / RCCL world-1 /
# from_pretrained), then BASELINE configs 2 and 4 and the default line with PIL output and the `parity` object.
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -x --durations=8 2>&1 | tail -25 | tee "$out/r02_pytest_gpu_b.log"
timeout 400 python bench.py --num-samples 1 --refs 1 --no-dialog --steps 3 --warmup 1 2> "$out/r02_bench_c2.err" \
    | tail -1 | tee "$out/r02_bench_c2_ns1_1ref.json" | cut -c1-400
timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2> "$out/r02_bench_ns16_pil.err" \
    | tail -1 | tee "$out/r02_bench_ns16_pil.json" | cut -c1-400
timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --output pt 2> "$out/r02_bench_ns16_pt.err" \
    | tail -1 | tee "$out/r02_bench_ns16_pt.json" | cut -c1-300
timeout 400 python tools/mixed_bench.py 2> "$out/r02_mixed.err" | tail -1 | tee "$out/r02_mixed_bucket_serving.json"
