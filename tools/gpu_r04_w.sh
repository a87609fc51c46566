#!/usr/bin/env bash
# Round 4, last experiment: conv_halo_kernel (8 x 16-pixel blocks, small batches) - tail tiles split four ways + patch offsets
# recomputed per slice instead of spilled.  Conv tests, tiny-UNet parity, then the A/B under the previous and the new build.
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
timeout 100 python -m pytest tests/test_gpu_ops.py -k "conv" tests/test_gpu_unet.py::test_unet_forward_vs_oracle tests/test_gpu_unet.py::test_unet_forward_any_latent_size tests/test_gpu_unet.py::test_unet_sdxl_small_batch_rows_are_position_independent -q -m gpu -p no:cacheprovider > "$out/r04_conv_small_tests.log" 2>&1
echo "pytest rc=$?"; tail -3 "$out/r04_conv_small_tests.log"
{ DIFFSENSEI_LIB=$PWD/diffsensei_amd/lib/libdiffsensei_hip_prev.so timeout 60 python tools/conv_small_ab.py 2 2>&1 | grep -v amdgpu.ids
  timeout 60 python tools/conv_small_ab.py 2 2>&1 | grep -v amdgpu.ids; } > "$out/r04_conv_small_ab.txt"
cat "$out/r04_conv_small_ab.txt"
