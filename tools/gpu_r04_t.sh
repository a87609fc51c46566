#!/usr/bin/env bash
# Round 4: the two remaining bench lines on the final tree (2048^2 with fp16 attention; round 2's operating point num_samples 16)
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
run() { # name, args...
  name=$1; shift
  timeout 280 python bench.py "$@" --no-cpu-baseline 2> "$out/$name.err" | tail -1 > "$out/$name.json"
  echo "$name rc=$? $(python -c "import json,sys; d=json.load(open('$out/$name.json')); print(d['value'], d['unet_forward'], d['roofline']['kernel'], d['roofline']['frac'])")"
}
run r04_bench_c5_2048_ns1_fp16_final --size 2048 --refs 4 --num-samples 1 --steps 2 --warmup 1
run r04_bench_ns16_final --num-samples 16 --steps 2 --warmup 1
