#!/usr/bin/env bash
# round 5, call C: gemm_pp_kernel build-time variants - EX (C stores allowed in flight under the next tile: 8 / 16) and the
# residual prefetch - correctness of each variant library (hand-over stress, fused-LN tests) and the batch-64 forward A/B.
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
L=$PWD/diffsensei_amd/lib
for v in ex16 pf ex16pf; do
  DIFFSENSEI_LIB=$L/libdiffsensei_hip_$v.so timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_ln_fusion.py -q -m gpu -p no:cacheprovider -k "gemm or ln or fused" > "$out/c_pytest_$v.log" 2>&1
  echo "pytest $v rc=$?"; tail -2 "$out/c_pytest_$v.log"
done
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -p no:cacheprovider -k "attention" > "$out/c_pytest_attn.log" 2>&1
echo "pytest attn rc=$?"; tail -2 "$out/c_pytest_attn.log"
for r in 1 2; do
  DIFFSENSEI_LIB=$L/libdiffsensei_hip_base.so AB_TAG=base timeout 300 python tools/forward_lib_ab.py 64 "$out/c_base_$r.json" 2>&1 | tail -1
  AB_TAG=new timeout 300 python tools/forward_lib_ab.py 64 "$out/c_new_$r.json" 2>&1 | tail -1
  for v in ex16 pf ex16pf; do
    DIFFSENSEI_LIB=$L/libdiffsensei_hip_$v.so AB_TAG=$v timeout 300 python tools/forward_lib_ab.py 64 "$out/c_${v}_$r.json" 2>&1 | tail -1
  done
done
python tools/forward_lib_ab.py --compare "$out"/c_new_*.json "$out"/c_ex16pf_*.json > "$out/r05_pp_ex16_pf_ab.txt"
python tools/forward_lib_ab.py --compare "$out"/c_base_*.json "$out"/c_new_*.json "$out"/c_ex16_*.json "$out"/c_pf_*.json "$out"/c_ex16pf_*.json | head -8
grep "gemm_pp\|self_attn" "$out/r05_pp_ex16_pf_ab.txt" | head -40
