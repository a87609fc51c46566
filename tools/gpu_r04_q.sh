#!/usr/bin/env bash
# Round 4: where does the fused LayerNorm pay (both kernel families, every UNet batch)?
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp


timeout 900 python tools/ln_fusion_sweep.py 2 4 8 16 32 64 2>&1 | grep -v amdgpu.ids > "$out/r04_ln_fusion_sweep.txt"
cat "$out/r04_ln_fusion_sweep.txt"
