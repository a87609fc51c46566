#!/usr/bin/env bash
# Round 4 evidence, part D: the whole 20-step trajectory of BASELINE configs[0] on the CPU oracle (nothing extrapolated) and the
# same call on the GPU (`parity` over all 20 steps + the decoded uint8 image), final tree
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
timeout 2400 python bench.py --steps 2 --warmup 1 --cpu-full > "$out/r04_bench_cpu_full.json" 2> "$out/r04_bench_cpu_full.err"
echo "rc=$?"; tail -1 "$out/r04_bench_cpu_full.json" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('value',d['value']); print('parity',d['parity']); print('cpu',d['cpu_baseline'])"
