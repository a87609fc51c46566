#!/usr/bin/env bash
# Round 4, visit F: fused LayerNorm A/B at the mid-size batches (partial fusion: norm2 / norm3 only where q|k is not on gemm_pp)
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
for b in 8 16 32; do
  timeout 600 python tools/ln_fusion_ab.py $b > "$out/r04_ln_fusion_ab_b$b.txt" 2>&1
  echo "ab $b rc=$?"; grep -E "UNet batch|LN fusion|layernorm|ln_finalize|gemm_pp" "$out/r04_ln_fusion_ab_b$b.txt"
done
