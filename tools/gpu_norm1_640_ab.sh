#!/usr/bin/env bash
# Round 6: norm1 of the 640-channel level folded into its GEMMs (the operand-swapped to_v consumer skips the 32-row pieces past its
# 640 output channels).  LayerNorm-fusion tests, the SDXL UNet parity tests, same-box forward A/B (library of the previous commit).
set -u
root="$GRAFT_REPO_ROOT"; out="$root/gpurun_out"; mkdir -p "$out/n1_ab"
cd "$root"
timeout 1200 python -m pytest tests/test_gpu_ln_fusion.py tests/test_gpu_unet.py tests/test_gpu_outlier_magnitudes.py -x -q > "$out/r06_norm1_640_tests.log" 2>&1
tail -4 "$out/r06_norm1_640_tests.log"
base="$root/diffsensei_amd/lib/libdiffsensei_hip_base.so"
for b in 64; do
  for rnd in 1 2; do
    DIFFSENSEI_LIB=$base AB_TAG=before timeout 600 python tools/forward_lib_ab.py $b "$out/n1_ab/before_b${b}_$rnd.json" 2>&1 | grep -v amdgpu.ids
    AB_TAG=norm1 timeout 600 python tools/forward_lib_ab.py $b "$out/n1_ab/norm1_b${b}_$rnd.json" 2>&1 | grep -v amdgpu.ids
  done
  { echo "=== UNet batch $b, 1024 x 1024: norm1 of the 640-channel level as a LayerNorm launch -> folded into proj_in / FF down-projection, q|k and the transposed to_v"; python tools/forward_lib_ab.py --compare "$out"/n1_ab/before_b${b}_*.json "$out"/n1_ab/norm1_b${b}_*.json; } > "$out/r06_norm1_640_ab_b$b.txt" 2>&1
  cat "$out/r06_norm1_640_ab_b$b.txt"
done
