#!/usr/bin/env bash
# Round 3, GPU call E: batched problems folded into the ping-pong GEMM: GEMM tests, per-shape timings of the hot shapes (the
# kernel's set_tile changed), level-1 / V^T A/B, RCCL + arena tests, the num_samples 32 bench line.
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_rccl.py -q -m gpu -x -k "gemm or rccl or broadcast" -s 2>&1 | grep -v "^\[transformers\]" | tail -8 | tee "$out/r03_e_pytest.log"
timeout 300 python tools/pp_epilogue_ab.py 2>&1 | grep -v "^\[transformers\]" | tail -11 | cut -c1-200 | tee "$out/r03_e_pp_shapes.txt"
timeout 300 python tools/l1_ab.py 2>&1 | tail -14 | tee "$out/r03_l1_vt_ab.txt"
timeout 600 python bench.py --num-samples 32 --steps 1 --warmup 1 --no-cpu-baseline > "$out/r03_e_bench_ns32.json" 2> "$out/r03_e_bench_ns32.err"
echo "bench ns32 rc=$?"; tail -1 "$out/r03_e_bench_ns32.json" | cut -c1-200
grep -a "unet_forward_ms_event_sum" "$out/r03_e_bench_ns32.err" | head -3
