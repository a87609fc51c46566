#!/usr/bin/env bash
# Round 6: 128 x 160 instantiation of gemm_t160_kernel (q|k projection of a batch-1 request: 16 x 16 = 256 blocks).
# Parity tests, then the in-situ plan A/B: gemm_t160 = 2 (64-row tiles only) vs 0 (rule picks 128-row tiles where they give one block per CU).
set -u
root="$GRAFT_REPO_ROOT"; out="$root/gpurun_out"; mkdir -p "$out"
cd "$root"
timeout 1200 python -m pytest tests/test_gpu_gemm_t160.py tests/test_gpu_gemm_g320.py tests/test_gpu_ln_fusion.py -x -q > "$out/r06_t160_tall_tests.log" 2>&1
tail -12 "$out/r06_t160_tall_tests.log"
for b in 2 4 8; do
  timeout 900 python tools/forward_plan_ab.py $b gemm_t160=2,0 2>&1 | grep -v amdgpu.ids > "$out/r06_t160_tall_forward_ab_b$b.txt"
  grep -v "^    [a-z]" "$out/r06_t160_tall_forward_ab_b$b.txt" | head -50
done
