#!/usr/bin/env bash
# Round 3, GPU call F: (1) the failing RCCL pipeline test with its full message + the new arena-rate test; (2) -ffast-math vs the
# reassociation-free flag set (lib/libdiffsensei_hip_strict.so): the bit-equality tests under the strict build, then the UNet
# forward event sum of both builds back to back, twice; (3) sp attention after the VALU trim vs the plain kernels.
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_rccl.py -q -m gpu -s 2>&1 | grep -v "^\[transformers\]" | tail -30 | tee "$out/r03_f_rccl.log"
S="$PWD/diffsensei_amd/lib/libdiffsensei_hip_strict.so"
DIFFSENSEI_LIB="$S" timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_unet.py -q -m gpu -x -k "not 1024 and not sdxl" 2>&1 | tail -4 | tee "$out/r03_f_strict_pytest.log"
for rnd in 1 2; do
  for which in fast strict; do
    if [ "$which" = strict ]; then export DIFFSENSEI_LIB="$S"; else unset DIFFSENSEI_LIB; fi
    timeout 600 python bench.py --num-samples 16 --steps 1 --warmup 1 --no-cpu-baseline > "$out/r03_f_bench_$which$rnd.json" 2> "$out/r03_f_bench_$which$rnd.err"
    echo "$which round $rnd: $(tail -1 $out/r03_f_bench_$which$rnd.json | cut -c70-100) $(grep -a unet_forward_ms_event_sum $out/r03_f_bench_$which$rnd.err | head -1)" | tee -a "$out/r03_fastmath_ab.txt"
  done
done
unset DIFFSENSEI_LIB
VARS=3,1,2 ROUNDS=5 timeout 300 python tools/attn_bench.py 2>&1 | tail -8 | tee "$out/r03_f_attn_bench.txt"
