#!/usr/bin/env bash
# round 5, call A: seamless-tile gemm_pp_kernel - correctness (GEMM / fused-LN / UNet tests, new large-shape tests) and the
# cross-library A/B of the batch-64 forward (base = round-4 library, new = this tree), interleaved rounds.
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > "$out/a_build.log" 2>&1
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_ln_fusion.py -q -m gpu -p no:cacheprovider -x -k "gemm or ln or fused or layernorm" > "$out/a_pytest_gemm.log" 2>&1
echo "pytest gemm rc=$?"; tail -5 "$out/a_pytest_gemm.log"
timeout 900 python -m pytest tests/test_gpu_unet.py -q -m gpu -p no:cacheprovider -x -k "not oracle" > "$out/a_pytest_unet.log" 2>&1
echo "pytest unet rc=$?"; tail -5 "$out/a_pytest_unet.log"
timeout 900 python -m pytest tests/test_gpu_large_shapes.py -q -m gpu -p no:cacheprovider -s -k "not oracle" > "$out/a_pytest_large.log" 2>&1
echo "pytest large rc=$?"; tail -25 "$out/a_pytest_large.log"
for r in 1 2; do
  DIFFSENSEI_LIB=$PWD/diffsensei_amd/lib/libdiffsensei_hip_base.so AB_TAG=base timeout 300 python tools/forward_lib_ab.py 64 "$out/ab_base_$r.json" 2>&1 | tail -1
  AB_TAG=new timeout 300 python tools/forward_lib_ab.py 64 "$out/ab_new_$r.json" 2>&1 | tail -1
done
python tools/forward_lib_ab.py --compare "$out"/ab_base_*.json "$out"/ab_new_*.json | tee "$out/r05_pp_seamless_ab.txt"
