#!/usr/bin/env bash
# Round 4: one-transcendental GELU + SGPR-base epilogue addresses - kernel tests, then the UNet forward under the previous
# library (HEAD 'ip_attn padding-key ... reverted', built as lib/libdiffsensei_hip_prev.so) and the new one, interleaved
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_ln_fusion.py tests/test_gpu_unet.py tests/test_gpu_encoders_true_shape.py -q -m gpu -p no:cacheprovider -x > "$out/r04_gelu_tests.log" 2>&1
echo "pytest rc=$?"; tail -5 "$out/r04_gelu_tests.log"
prev=$PWD/diffsensei_amd/lib/libdiffsensei_hip_prev.so
for rnd in 1 2; do
  for which in prev new; do
    if [ $which = prev ]; then export DIFFSENSEI_LIB=$prev; else unset DIFFSENSEI_LIB; fi
    timeout 600 python tools/forward_env_ab.py 64 DIFFSENSEI_NO_SUCH_SWITCH > "$out/r04_gelu_ab_b64_${which}_$rnd.txt" 2>&1
    echo "b64 $which $rnd rc=$?"; grep -i "forward\|event sum" "$out/r04_gelu_ab_b64_${which}_$rnd.txt" | tail -4
  done
done
for which in prev new; do
  if [ $which = prev ]; then export DIFFSENSEI_LIB=$prev; else unset DIFFSENSEI_LIB; fi
  timeout 300 python tools/forward_env_ab.py 2 DIFFSENSEI_NO_SUCH_SWITCH > "$out/r04_gelu_ab_b2_${which}.txt" 2>&1
  echo "b2 $which rc=$?"; grep -i "forward\|event sum" "$out/r04_gelu_ab_b2_${which}.txt" | tail -4
done
