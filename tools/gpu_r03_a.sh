#!/usr/bin/env bash
# Round 3, GPU call A: new parity tests (1024x1024 SDXL forward vs oracle + batch-32 rows, true-shape encoders, __call__ -> PIL
# vs the whole-call oracle), first contact of the experimental 4-wave GEMM, one short bench run through the new parity path.
# Library: built with --experimental BEFORE the call (python -m diffsensei_amd.build --experimental --force).
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_call_parity.py tests/test_gpu_encoders_true_shape.py \
    "tests/test_gpu_unet.py::test_unet_sdxl_forward_vs_oracle_1024" -q -m gpu -s -x 2>&1 | tail -25 | tee "$out/r03_a_pytest.log"
timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "w4" 2>&1 | tail -3 | tee "$out/r03_a_w4_pytest.log"
timeout 300 python tools/w4_check.py 2> "$out/r03_w4_check.err" | tee "$out/r03_w4_check.txt"
tail -3 "$out/r03_w4_check.err"
timeout 600 python bench.py --steps 1 --warmup 1 > "$out/r03_a_bench.json" 2> "$out/r03_a_bench.err"
echo "bench rc=$?"
tail -1 "$out/r03_a_bench.json" | cut -c1-400
grep -a "parity" "$out/r03_a_bench.err" | tail -2
tail -5 "$out/r03_a_bench.err"
