#!/usr/bin/env bash
# End of round 6: the plain-epilogue form of gemm_g320_kernel for q|k at M = 8192, N = 2560 (UNet batch 8 at 1024^2, batch 2 at 2048^2).
# Parity tests, then the in-situ A/B through the library option gemm_g320 (1 = rule off) at both shapes (and batch 2 at 1024^2: GEGLU form).
set -u
root="$GRAFT_REPO_ROOT"; out="$root/gpurun_out"; mkdir -p "$out"
cd "$root"
timeout 1200 python -m pytest tests/test_gpu_gemm_g320.py tests/test_gpu_gemm_t160.py tests/test_gpu_ln_fusion.py -x -q > "$out/r06_g320_plain_tests.log" 2>&1
tail -4 "$out/r06_g320_plain_tests.log"
timeout 900 python tools/forward_plan_ab.py 8 gemm_g320=1,0 2>&1 | grep -v amdgpu.ids > "$out/r06_g320_plain_forward_ab_b8.txt"
grep -v "^    [a-z]" "$out/r06_g320_plain_forward_ab_b8.txt" | head -24
timeout 900 python tools/forward_plan_ab.py 2 gemm_g320=1,0 256 2>&1 | grep -v amdgpu.ids > "$out/r06_g320_plain_forward_ab_b2_2048.txt"
grep -v "^    [a-z]" "$out/r06_g320_plain_forward_ab_b2_2048.txt" | head -24
