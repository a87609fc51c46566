#!/usr/bin/env bash
# Round 6: ip_attn_kernel<4,false,true> - the work of the padding keys 80..95 (a sixth of the scores) never issued.
# Parity tests, back-to-back microbenchmark (ip_attn_variant 3 = the same kernel without the specialisation), in-situ option A/B.
set -u
root="$GRAFT_REPO_ROOT"; out="$root/gpurun_out"; mkdir -p "$out"
cd "$root"
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "masked_ip or processors or region" > "$out/r06_ipattn_t16_tests.log" 2>&1
tail -3 "$out/r06_ipattn_t16_tests.log"
{
for shape in "64 20 32 32" "64 10 64 64" "8 20 32 32" "2 20 32 32"; do
  for rnd in 1 2; do
    echo -n "all 96 key slots   : "; DS_OPTIONS=ip_attn_variant=3 python tools/one_ipattn.py $shape 20 2>&1 | grep -v amdgpu.ids
    echo -n "keys 80..95 skipped: "; python tools/one_ipattn.py $shape 20 2>&1 | grep -v amdgpu.ids
  done
done
} > "$out/r06_ipattn_t16_microbench.txt" 2>&1
cat "$out/r06_ipattn_t16_microbench.txt"
for b in 64 2; do
  timeout 900 python tools/forward_option_ab.py $b ip_attn_variant=3,0 2>&1 | grep -v amdgpu.ids > "$out/r06_ipattn_t16_option_ab_b$b.txt"
  cat "$out/r06_ipattn_t16_option_ab_b$b.txt" | head -20
done
