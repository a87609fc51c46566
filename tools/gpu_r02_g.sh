#!/usr/bin/env bash
# Round 2, GPU call G: re-check of the odd-size UNet test after the oracle fix, ip_attn query-tile sweep at batch 32,
# gemm_pp (even-rounds grid) vs the automatic dispatch on the num_samples 2 / 4 / 8 shapes.
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_unet.py tests/test_gpu_attention_fp8.py -q -m gpu -k "any_latent_size or fp8" 2>&1 | tail -5 | tee "$out/r02_pytest_g.log"
timeout 200 python tools/ipattn_bench.py 2>&1 | grep -v amdgpu.ids | tee "$out/r02_ipattn_bench_b32.txt"
for b in 4 8 16; do
echo "== batch $b" | tee -a "$out/r02_gemm_pp_dispatch_sweep.txt"
timeout 300 python tools/gemm_bench.py --variants 0,3 --batch $b --reps 20 2>&1 | grep -v "amdgpu.ids\|^conv" | tee -a "$out/r02_gemm_pp_dispatch_sweep.txt"
done
