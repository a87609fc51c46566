#!/usr/bin/env bash
# Round 6: gemm_t160_kernel requests its epilogue operands (bias, residual rows, LayerNorm partial sums, c) ahead of its first
# DMA piece.  Its parity tests, then a same-box forward A/B at UNet batch 2 (library of the previous commit) - three rounds.
set -u
root="$GRAFT_REPO_ROOT"; out="$root/gpurun_out"; mkdir -p "$out/t160_ab"
cd "$root"
timeout 1200 python -m pytest tests/test_gpu_gemm_t160.py tests/test_gpu_ln_fusion.py tests/test_gpu_unet.py -x -q > "$out/r06_t160_prefetch_tests.log" 2>&1
tail -4 "$out/r06_t160_prefetch_tests.log"
base="$root/diffsensei_amd/lib/libdiffsensei_hip_base.so"
for rnd in 1 2 3; do
  DIFFSENSEI_LIB=$base AB_TAG=before timeout 600 python tools/forward_lib_ab.py 2 "$out/t160_ab/before_$rnd.json" 2>&1 | grep -v amdgpu.ids
  AB_TAG=early timeout 600 python tools/forward_lib_ab.py 2 "$out/t160_ab/early_$rnd.json" 2>&1 | grep -v amdgpu.ids
done
{ echo "=== UNet batch 2, 1024 x 1024: gemm_t160_kernel's epilogue operands requested in the epilogue -> ahead of the first DMA piece"; python tools/forward_lib_ab.py --compare "$out"/t160_ab/before_*.json "$out"/t160_ab/early_*.json; } > "$out/r06_t160_prefetch_ab_b2.txt" 2>&1
cat "$out/r06_t160_prefetch_ab_b2.txt"
