#!/usr/bin/env bash
# round 5, call G: the ring-buffered small-grid GEMM with a k-step read-ahead (asm LDS-DMA, counted LDS waits): tests + batch-2 / batch-8 forward A/B
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
python - <<'PY' || exit 3
from diffsensei_amd import build
import os
assert open(os.path.join(build.LIBDIR, "build.stamp")).read().strip() == build._digest(), "sources changed after the library was built"
PY
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_ln_fusion.py -q -m gpu -p no:cacheprovider -k "gemm or ln or fused" > "$out/g_pytest_gemm.log" 2>&1
echo "pytest gemm rc=$?"; tail -2 "$out/g_pytest_gemm.log"
timeout 600 python -m pytest tests/test_gpu_unet.py -q -m gpu -p no:cacheprovider -k "not oracle" > "$out/g_pytest_unet.log" 2>&1
echo "pytest unet rc=$?"; tail -2 "$out/g_pytest_unet.log"
L=$PWD/diffsensei_amd/lib
for b in 2 8; do
  for r in 1 2; do
    DIFFSENSEI_LIB=$L/libdiffsensei_hip_base.so AB_TAG=base timeout 300 python tools/forward_lib_ab.py $b "$out/g_base_${b}_$r.json" 2>&1 | tail -1
    AB_TAG=new timeout 300 python tools/forward_lib_ab.py $b "$out/g_new_${b}_$r.json" 2>&1 | tail -1
  done
  python tools/forward_lib_ab.py --compare "$out"/g_base_${b}_*.json "$out"/g_new_${b}_*.json > "$out/r05_ring_readahead_ab_b$b.txt"
  head -16 "$out/r05_ring_readahead_ab_b$b.txt"
done
