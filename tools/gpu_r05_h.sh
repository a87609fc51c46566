#!/usr/bin/env bash
# round 5, call H: gemm_pp_kernel's branch-free epilogues without the LDS transposition (v_permlane32_swap + row-per-lane 16-byte
# stores) vs the LDS transposition (build variant ldsepi): correctness, then the batch-64 forward A/B
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
python - <<'PY' || exit 3
from diffsensei_amd import build
import os
assert open(os.path.join(build.LIBDIR, "build.stamp")).read().strip() == build._digest(), "sources changed after the library was built"
PY
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_ln_fusion.py tests/test_gpu_vae.py -q -m gpu -p no:cacheprovider -x > "$out/h_pytest_ops.log" 2>&1
echo "pytest ops rc=$?"; tail -4 "$out/h_pytest_ops.log"
timeout 900 python -m pytest tests/test_gpu_unet.py tests/test_gpu_large_shapes.py tests/test_gpu_pipeline.py tests/test_gpu_call_parity.py -q -m gpu -p no:cacheprovider -x -k "not oracle_1536 and not oracle_2048" > "$out/h_pytest_unet.log" 2>&1
echo "pytest unet rc=$?"; tail -4 "$out/h_pytest_unet.log"
L=$PWD/diffsensei_amd/lib
for r in 1 2; do
  DIFFSENSEI_LIB=$L/libdiffsensei_hip_ldsepi.so AB_TAG=ldsepi timeout 300 python tools/forward_lib_ab.py 64 "$out/h_lds_$r.json" 2>&1 | tail -1
  AB_TAG=direct timeout 300 python tools/forward_lib_ab.py 64 "$out/h_direct_$r.json" 2>&1 | tail -1
done
python tools/forward_lib_ab.py --compare "$out"/h_lds_*.json "$out"/h_direct_*.json > "$out/r05_pp_direct_epilogue_ab.txt"
head -34 "$out/r05_pp_direct_epilogue_ab.txt"
