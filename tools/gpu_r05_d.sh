#!/usr/bin/env bash
# round 5, call D: conv_halo256 fragment read-ahead, packed residual adds / dot-product statistics, EX = 16 - correctness + A/B
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
python - <<'PY' || exit 3
from diffsensei_amd import build
import os
st = open(os.path.join(build.LIBDIR, "build.stamp")).read().strip()
assert st == build._digest(), "sources changed after the library was built: snapshot taken mid-edit"
print("tree matches the built library")
PY
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_ln_fusion.py tests/test_gpu_vae.py -q -m gpu -p no:cacheprovider -x > "$out/d_pytest_ops.log" 2>&1
echo "pytest ops rc=$?"; tail -4 "$out/d_pytest_ops.log"
timeout 900 python -m pytest tests/test_gpu_unet.py tests/test_gpu_large_shapes.py -q -m gpu -p no:cacheprovider -x -k "not oracle" > "$out/d_pytest_unet.log" 2>&1
echo "pytest unet rc=$?"; tail -4 "$out/d_pytest_unet.log"
L=$PWD/diffsensei_amd/lib
for r in 1 2; do
  DIFFSENSEI_LIB=$L/libdiffsensei_hip_base.so AB_TAG=base timeout 300 python tools/forward_lib_ab.py 64 "$out/d_base_$r.json" 2>&1 | tail -1
  AB_TAG=new timeout 300 python tools/forward_lib_ab.py 64 "$out/d_new_$r.json" 2>&1 | tail -1
done
python tools/forward_lib_ab.py --compare "$out"/d_base_*.json "$out"/d_new_*.json > "$out/r05_forward_ab_d.txt"
head -32 "$out/r05_forward_ab_d.txt"
