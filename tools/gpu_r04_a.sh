#!/usr/bin/env bash
# Round 4, visit A: (1) gemm_pp with every operand load hitting L2 (ablation build) next to the production stream, (2) the new
# parity tests at the benched batch.
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
DIFFSENSEI_LIB=$PWD/diffsensei_amd/lib/libdiffsensei_hip_ablation.so timeout 300 python tools/pp_l2hit_ab.py 0,32 > "$out/r04_pp_l2hit_ab.txt" 2>&1
echo "l2hit rc=$?"; cat "$out/r04_pp_l2hit_ab.txt"
timeout 1500 python -m pytest -q -m gpu -p no:cacheprovider --durations=8 -s \
   tests/test_gpu_vae.py::test_vae_decode_1024_vs_oracle tests/test_gpu_vae.py::test_wide_attention_f16_at_decode_size \
   tests/test_gpu_pipeline_variants.py::test_callback_may_return_replaced_latents tests/test_gpu_rccl.py \
   tests/test_gpu_unet.py::test_unet_sdxl_forward_vs_oracle_1024 > "$out/r04_pytest_new.log" 2>&1
echo "pytest rc=$?"; grep -v "^$" "$out/r04_pytest_new.log" | tail -40
