#!/usr/bin/env bash
# Round 6: gemm_pp_kernel's branch-free epilogue on whole 64-column strips of a ragged last tile column (N = 640).
# GPU tests of the change, then a same-box A/B of the UNet forward on the library before the change (lib/libdiffsensei_hip_base.so,
# built from the previous commit) and the current one, two interleaved rounds at UNet batch 64 and one at batch 8 / 2.
set -u
root="$GRAFT_REPO_ROOT"; out="$root/gpurun_out"; mkdir -p "$out/strips_ab"
cd "$root"
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_ln_fusion.py -x -q -k "gemm or producer or consumer or fused or chain" > "$out/r06_pp_strips_tests.log" 2>&1
tail -4 "$out/r06_pp_strips_tests.log"
base="$root/diffsensei_amd/lib/libdiffsensei_hip_base.so"
for b in 64 8 2; do
  rounds="1 2"; [ $b != 64 ] && rounds="1"
  for rnd in $rounds; do
    DIFFSENSEI_LIB=$base AB_TAG=base timeout 600 python tools/forward_lib_ab.py $b "$out/strips_ab/base_b${b}_$rnd.json" 2>&1 | grep -v amdgpu.ids
    AB_TAG=strips timeout 600 python tools/forward_lib_ab.py $b "$out/strips_ab/strips_b${b}_$rnd.json" 2>&1 | grep -v amdgpu.ids
  done
  { echo "=== UNet batch $b, 1024 x 1024: library before the change -> with whole-strip epilogues on ragged tile columns"; python tools/forward_lib_ab.py --compare "$out"/strips_ab/base_b${b}_*.json "$out"/strips_ab/strips_b${b}_*.json; } > "$out/r06_pp_strips_ab_b$b.txt" 2>&1
done
head -40 "$out/r06_pp_strips_ab_b64.txt"
