#!/usr/bin/env bash
# Round 2, GPU call R: the new tile hand-over parity test, then the whole-call parity run on the final tree: all 20 steps of
# BASELINE configs[0] + the fp32 VAE decode on the host oracle vs the GPU engine (latents and decoded image).
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "handover" 2>&1 | tail -3
timeout 1500 python bench.py --steps 2 --warmup 1 --cpu-full > "$out/r02_bench_ns16_cpu_full_final.json" 2> "$out/r02_bench_ns16_cpu_full_final.err"
echo "cpu-full rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r02_bench_ns16_cpu_full_final.json").read().strip().splitlines()[-1])
print("value %.4f panels/s; parity %s; cpu_baseline %s" % (d["value"], d["parity"], d["cpu_baseline"]))
PY
