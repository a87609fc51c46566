#!/usr/bin/env bash
# Round 6: 16-byte O stores (v_permlane32_swap hand-over, as measured on ip_attn_kernel) in self_attn_sp_kernel and
# self_attn_kernel<1|2>.  base = HEAD before the change (lib/libdiffsensei_hip_base.so), default = with it.
# Parity tests, back-to-back microbenchmark (tools/attn_bench.py per library, interleaved), in-situ forward A/B at batch 64 / 8 / 2.
set -u
root="$GRAFT_REPO_ROOT"; out="$root/gpurun_out"; mkdir -p "$out/at_ab"
cd "$root"
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_outlier_magnitudes.py -x -q -k "attention or attn or outlier" > "$out/r06_attn_store_tests.log" 2>&1
tail -3 "$out/r06_attn_store_tests.log"
base="$root/diffsensei_amd/lib/libdiffsensei_hip_base.so"
{
for rnd in 1 2; do
  echo "--- 8-byte O stores (before), round $rnd"; DIFFSENSEI_LIB=$base ROUNDS=3 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids
  echo "--- 16-byte O stores, round $rnd";         ROUNDS=3 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids
done
} > "$out/r06_attn_store_microbench.txt" 2>&1
cat "$out/r06_attn_store_microbench.txt"
for b in 64 2 8; do
  for rnd in 1 2; do
    DIFFSENSEI_LIB=$base AB_TAG=before timeout 600 python tools/forward_lib_ab.py $b "$out/at_ab/before_b${b}_$rnd.json" 2>&1 | grep -v amdgpu.ids
    AB_TAG=new timeout 600 python tools/forward_lib_ab.py $b "$out/at_ab/new_b${b}_$rnd.json" 2>&1 | grep -v amdgpu.ids
  done
  { echo "=== UNet batch $b, 1024 x 1024: self-attention O as eight 8-byte stores per row and lane -> four 16-byte stores"; python tools/forward_lib_ab.py --compare "$out"/at_ab/before_b${b}_*.json "$out"/at_ab/new_b${b}_*.json; } > "$out/r06_attn_store_forward_ab_b$b.txt" 2>&1
  head -14 "$out/r06_attn_store_forward_ab_b$b.txt"
done
