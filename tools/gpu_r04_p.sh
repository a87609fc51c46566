#!/usr/bin/env bash
# Round 4: fused LayerNorm on the 128-wide kernels (small batches, 640-channel level) + the GELU / epilogue-address changes with
# the rounding pinned: kernel and UNet tests, position independence, in-situ A/B
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_ln_fusion.py tests/test_gpu_ops.py tests/test_gpu_unet.py tests/test_gpu_encoders_true_shape.py -q -m gpu -p no:cacheprovider > "$out/r04_wide_ln_tests.log" 2>&1
echo "pytest rc=$?"; tail -12 "$out/r04_wide_ln_tests.log"
: > "$out/r04_determinism_after_fix.txt"
for args in "96 8 2" "96 2 2" "72 3 2"; do
  timeout 300 python tools/replicate_determinism.py $args 2>&1 | grep -v amdgpu.ids >> "$out/r04_determinism_after_fix.txt"
done
cat "$out/r04_determinism_after_fix.txt"
for b in 2 8; do
  timeout 400 python tools/forward_env_ab.py $b DIFFSENSEI_LN_FUSION > "$out/r04_wide_ln_ab_b$b.txt" 2>&1
  echo "b$b rc=$?"; grep -v amdgpu.ids "$out/r04_wide_ln_ab_b$b.txt" | head -24
done
prev=$PWD/diffsensei_amd/lib/libdiffsensei_hip_prev.so
for which in prev new; do
  if [ $which = prev ]; then export DIFFSENSEI_LIB=$prev; else unset DIFFSENSEI_LIB; fi
  timeout 600 python tools/forward_env_ab.py 64 DIFFSENSEI_NO_SUCH_SWITCH > "$out/r04_wide_ln_ab_b64_$which.txt" 2>&1
  echo "b64 $which rc=$?"; grep -A12 "SWITCH=0: forward" "$out/r04_wide_ln_ab_b64_$which.txt"
done
