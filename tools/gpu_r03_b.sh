#!/usr/bin/env bash
# Round 3, GPU call B: VAE decoder in scaled fp16 (tests + the bench parity object with uint8 statistics), first contact of
# the experimental 4-wave GEMM (the library must be the --experimental build; DIFFSENSEI_BUILD_EXPERIMENTAL keeps it so).
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
export DIFFSENSEI_BUILD_EXPERIMENTAL=1
timeout 600 python -m pytest tests/test_gpu_vae.py tests/test_gpu_call_parity.py -q -m gpu -s -x 2>&1 | grep -v "^\[transformers\]" | tail -25 | tee "$out/r03_b_pytest.log"
timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "w4" 2>&1 | tail -15 | tee "$out/r03_b_w4_pytest.log"
timeout 300 python tools/w4_check.py 2> "$out/r03_w4_check.err" | tee "$out/r03_w4_check.txt"
tail -3 "$out/r03_w4_check.err"
timeout 300 python tools/vae_bench.py 2>&1 | tail -8 | tee "$out/r03_b_vae_bench.txt"
timeout 600 python bench.py --steps 1 --warmup 1 --no-roofline > "$out/r03_b_bench.json" 2> "$out/r03_b_bench.err"
echo "bench rc=$?"
tail -1 "$out/r03_b_bench.json" | cut -c1-300
grep -a "^parity" "$out/r03_b_bench.err" | tail -1
