#!/usr/bin/env bash
# Round 4: which change makes rows of a replicated batch differ (tests/test_gpu_unet.py::test_unet_sdxl_partial_layernorm_fusion_768)?
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
L=$PWD/diffsensei_amd/lib
: > "$out/r04_determinism_bisect.txt"
for lib in prev new varA varB; do
  if [ $lib = new ]; then unset DIFFSENSEI_LIB; else export DIFFSENSEI_LIB=$L/libdiffsensei_hip_$lib.so; fi
  timeout 300 python tools/replicate_determinism.py 96 8 3 2>&1 | grep -v amdgpu.ids >> "$out/r04_determinism_bisect.txt"
done
unset DIFFSENSEI_LIB
DIFFSENSEI_LN_FUSION=0 timeout 300 python tools/replicate_determinism.py 96 8 2 2>&1 | grep -v amdgpu.ids >> "$out/r04_determinism_bisect.txt"
timeout 300 python tools/replicate_determinism.py 128 4 2 2>&1 | grep -v amdgpu.ids >> "$out/r04_determinism_bisect.txt"
cat "$out/r04_determinism_bisect.txt"
