#!/usr/bin/env bash
# Round 6, ip_attn_kernel: (a) every LDS fragment of a phase requested ahead of the phase's MFMAs, (b) the panel staging's global
# loads (and the box load) requested in one batch, (c) 16-byte O stores (v_permlane32_swap).  Libraries: base = before all of it,
# st8 = (a) + (b) with the old 8-byte stores, default = all three.  Parity tests, back-to-back microbenchmark, in-situ forward A/B.
set -u
root="$GRAFT_REPO_ROOT"; out="$root/gpurun_out"; mkdir -p "$out/ip_ab2"
cd "$root"
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "masked_ip or processors or region" > "$out/r06_ipattn_tests.log" 2>&1
tail -3 "$out/r06_ipattn_tests.log"
base="$root/diffsensei_amd/lib/libdiffsensei_hip_base.so"; st8="$root/diffsensei_amd/lib/libdiffsensei_hip_st8.so"
{
for shape in "64 20 32 32" "64 10 64 64" "8 20 32 32" "2 20 32 32" "2 10 128 128"; do
  for rnd in 1 2; do
    echo -n "before              : "; DIFFSENSEI_LIB=$base python tools/one_ipattn.py $shape 20 2>&1 | grep -v amdgpu.ids
    echo -n "batched staging     : "; DIFFSENSEI_LIB=$st8 python tools/one_ipattn.py $shape 20 2>&1 | grep -v amdgpu.ids
    echo -n " + 16-byte O stores : "; python tools/one_ipattn.py $shape 20 2>&1 | grep -v amdgpu.ids
  done
  for mb in 512 2048; do
    echo -n "   min_blocks $mb : "; DS_OPTIONS=ip_attn_min_blocks=$mb python tools/one_ipattn.py $shape 20 2>&1 | grep -v amdgpu.ids
  done
  echo -n "   ring variant   : "; DS_OPTIONS=ip_attn_variant=2 python tools/one_ipattn.py $shape 20 2>&1 | grep -v amdgpu.ids
done
} > "$out/r06_ipattn_microbench2.txt" 2>&1
cat "$out/r06_ipattn_microbench2.txt"
for b in 64 2 8; do
  for rnd in 1 2; do
    DIFFSENSEI_LIB=$base AB_TAG=before timeout 600 python tools/forward_lib_ab.py $b "$out/ip_ab2/before_b${b}_$rnd.json" 2>&1 | grep -v amdgpu.ids
    AB_TAG=new timeout 600 python tools/forward_lib_ab.py $b "$out/ip_ab2/new_b${b}_$rnd.json" 2>&1 | grep -v amdgpu.ids
  done
  { echo "=== UNet batch $b, 1024 x 1024: ip_attn_kernel of round 5 -> fragments a phase ahead, batched panel staging, 16-byte O stores"; python tools/forward_lib_ab.py --compare "$out"/ip_ab2/before_b${b}_*.json "$out"/ip_ab2/new_b${b}_*.json; } > "$out/r06_ipattn_forward_ab_b$b.txt" 2>&1
  head -14 "$out/r06_ipattn_forward_ab_b$b.txt"
done
