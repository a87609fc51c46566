#!/usr/bin/env bash
# Round 3, GPU call D: software-pipelined attention as the automatic choice for large grids: its tests, the 1024x1024 SDXL
# parity test with it in the plan, attention microbench, the default bench line, and the num_samples 32 line.
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "self_attention or AttnProcessor or attn" 2>&1 | tail -5 | tee "$out/r03_d_pytest_attn.log"
timeout 900 python -m pytest "tests/test_gpu_unet.py::test_unet_sdxl_forward_vs_oracle_1024" -q -m gpu -s 2>&1 | grep -v "^\[transformers\]" | tail -6 | tee "$out/r03_d_pytest_1024.log"
ROUNDS=5 timeout 300 python tools/attn_bench.py 2>&1 | tail -8 | tee "$out/r03_self_attn_sp.txt"
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$out/r03_d_bench_ns16.json" 2> "$out/r03_d_bench_ns16.err"
echo "bench ns16 rc=$?"; tail -1 "$out/r03_d_bench_ns16.json" | cut -c1-200
grep -a "unet_forward_ms_event_sum\|self_attn" "$out/r03_d_bench_ns16.err" | head -6
timeout 600 python bench.py --num-samples 32 --steps 1 --warmup 1 --no-cpu-baseline > "$out/r03_d_bench_ns32.json" 2> "$out/r03_d_bench_ns32.err"
echo "bench ns32 rc=$?"; tail -1 "$out/r03_d_bench_ns32.json" | cut -c1-200
grep -a "unet_forward_ms_event_sum" "$out/r03_d_bench_ns32.err" | head -3
