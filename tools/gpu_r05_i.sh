#!/usr/bin/env bash
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
python - <<'PY' || exit 3
from diffsensei_amd import build
import os
assert open(os.path.join(build.LIBDIR, "build.stamp")).read().strip() == build._digest(), "sources changed after the library was built"
PY
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_large_shapes.py -q -m gpu -p no:cacheprovider -k "attention or attn or processors" > "$out/i_pytest_attn.log" 2>&1
echo "pytest attn rc=$?"; tail -3 "$out/i_pytest_attn.log"
L=$PWD/diffsensei_amd/lib
DIFFSENSEI_LIB=$L/libdiffsensei_hip_nopersist.so AB_TAG=nopersist timeout 400 python tools/attn_lib_ab.py "$out/i_attn_nopersist.json" 2>&1 | grep -v amdgpu.ids | cut -c1-120
AB_TAG=persist timeout 400 python tools/attn_lib_ab.py "$out/i_attn_persist.json" 2>&1 | grep -v amdgpu.ids | cut -c1-120
for r in 1 2; do
  DIFFSENSEI_LIB=$L/libdiffsensei_hip_nopersist.so AB_TAG=nopersist timeout 300 python tools/forward_lib_ab.py 64 "$out/i_np_$r.json" 2>&1 | tail -1
  AB_TAG=persist timeout 300 python tools/forward_lib_ab.py 64 "$out/i_p_$r.json" 2>&1 | tail -1
done
python tools/forward_lib_ab.py --compare "$out"/i_np_*.json "$out"/i_p_*.json > "$out/r05_self_attn_persistent_ab.txt"
head -12 "$out/r05_self_attn_persistent_ab.txt"
python tools/attn_lib_ab.py --compare "$out/i_attn_nopersist.json" "$out/i_attn_persist.json" >> "$out/r05_self_attn_persistent_ab.txt"
