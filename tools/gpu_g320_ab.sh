#!/usr/bin/env bash
# Round 6: gemm_g320_kernel (256 x 320 GEGLU tiles, one block per CU) for the feed-forward projection of a batch-1 request.
# Parity tests, back-to-back microbenchmark against the 128-packed kernels, in-situ forward A/B (plan switch gemm_g320).
set -u
root="$GRAFT_REPO_ROOT"; out="$root/gpurun_out"; mkdir -p "$out"
cd "$root"
timeout 900 python -m pytest tests/test_gpu_gemm_g320.py -x -q > "$out/r06_g320_tests.log" 2>&1
tail -15 "$out/r06_g320_tests.log"
{
  timeout 300 python tools/one_g320.py 2048 10240 1280 20
  timeout 300 python tools/one_g320.py 4096 10240 1280 20
  timeout 300 python tools/one_g320.py 8192 5120 640 20
} 2>&1 | grep -v amdgpu.ids > "$out/r06_g320_microbench.txt"
cat "$out/r06_g320_microbench.txt"
for b in 2; do
  timeout 900 python tools/forward_plan_ab.py $b gemm_g320=1,0 2>&1 | grep -v amdgpu.ids > "$out/r06_g320_forward_ab_b$b.txt"
  head -40 "$out/r06_g320_forward_ab_b$b.txt"
done
