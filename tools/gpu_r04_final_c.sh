#!/usr/bin/env bash
# Round 4 evidence, part C: the other BASELINE.json configs on ONE GPU (whole __call__, prompt -> PIL), final tree
# (FULL=1 adds the fp16-attention 2048^2 line and the num_samples-16 line)
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
if [[ -n "${RETEST:-}" ]]; then
  timeout 600 python -m pytest $RETEST -q -m gpu -p no:cacheprovider > "$out/r04_pytest_gpu_retest.log" 2>&1
  echo "retest rc=$?"; tail -3 "$out/r04_pytest_gpu_retest.log"
fi
run() { # name, args...
  name=$1; shift
  timeout 700 python bench.py "$@" --no-cpu-baseline 2> "$out/$name.err" | tail -1 > "$out/$name.json"
  echo "$name rc=$? $(python -c "import json,sys; d=json.load(open('$out/$name.json')); print(d['value'], d['unet_forward'], d['roofline']['kernel'], d['roofline']['frac'])")"
}
run r04_bench_c2_ns1_1ref_final --num-samples 1 --refs 1 --no-dialog --steps 3 --warmup 1
run r04_bench_c3_mllm_ns4_final --mllm --num-samples 4 --steps 2 --warmup 1
timeout 600 python tools/mixed_bench.py 2>/dev/null | tail -1 | tee "$out/r04_mixed_bucket_serving.json" | cut -c1-300
run r04_bench_c5_2048_ns1_fp8_final --size 2048 --refs 4 --num-samples 1 --attn fp8 --steps 2 --warmup 1
if [[ -n "${FULL:-}" ]]; then
  run r04_bench_c5_2048_ns1_fp16_final --size 2048 --refs 4 --num-samples 1 --steps 2 --warmup 1
  run r04_bench_ns16_final --num-samples 16 --steps 2 --warmup 1
fi
