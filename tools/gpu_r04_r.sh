#!/usr/bin/env bash
# Round 4: norm1 fused on the 128-wide kernels as well (operand-swapped consumer) - kernel tests, position independence at small
# batches, the fusion sweep with the default rule
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ln_fusion.py -q -m gpu -p no:cacheprovider > "$out/r04_wide_ln_tests3.log" 2>&1
echo "pytest rc=$?"; tail -6 "$out/r04_wide_ln_tests3.log"
: > "$out/r04_determinism_small_batch.txt"
for args in "96 2 2" "72 3 2" "128 1 2"; do
  timeout 300 python tools/replicate_determinism.py $args 2>&1 | grep -v amdgpu.ids >> "$out/r04_determinism_small_batch.txt"
done
cat "$out/r04_determinism_small_batch.txt"
timeout 900 python tools/ln_fusion_sweep.py 2 4 8 16 32 64 2>&1 | grep -v amdgpu.ids > "$out/r04_ln_fusion_sweep2.txt"
cat "$out/r04_ln_fusion_sweep2.txt"
