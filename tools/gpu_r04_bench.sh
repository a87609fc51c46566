#!/usr/bin/env bash
# Round 4: headline bench on the current tree (driver defaults apart from fewer timed calls)
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
tag="${1:-mid}"
timeout 1500 python bench.py --steps 2 --warmup 1 > "$out/r04_bench_default_ns32_$tag.json" 2> "$out/r04_bench_default_ns32_$tag.err"
echo "bench rc=$?"
tail -1 "$out/r04_bench_default_ns32_$tag.json" | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('value',d['value'],'ms_per_step',d['ms_per_step'],'frac',d['roofline']['frac'],'avg_launch_us',d['roofline']['avg_launch_us'])
print('forward',d['unet_forward'])
print('parity',d['parity'])
print('cpu',d['cpu_baseline']['value'], d['config'].get('vae_precision'))
print('launches/step', d['config']['kernel_launches_per_denoise_step'])
"
grep -A80 '"per_kernel"' "$out/r04_bench_default_ns32_$tag.err" | grep -E '^  "|"ms"' | paste - - | head -16
